#!/bin/bash
# start-up stagger sweep of conv_hpipe (SYLPH_HPIPE_STAGGER_US) on the 3x3 layer shapes
for us in 0 6 12 20 30; do echo "== stagger $us us"; SYLPH_HPIPE_STAGGER_US=$us python tools/bench_3x3.py 64 20; done
