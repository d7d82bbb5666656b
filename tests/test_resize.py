"""Input pipeline: the oracle's restatement of Pillow's 8-bit BILINEAR resampling against outputs of the real
Pillow (tests/golden/g9_resize.npz) -- bit exact -- and ResizeShortestEdge shapes.  CPU only."""
import os

import numpy as np
import pytest

from oracle import resize as R


@pytest.fixture(scope="module")
def g9(golden_dir):
    return np.load(os.path.join(golden_dir, "g9_resize.npz"))


def test_oracle_resize_is_bit_exact_with_pillow(g9):
    for i, (h, w, nh, nw) in enumerate(g9["cases"]):
        got = R.pil_bilinear_resize(g9[f"in{i}"], int(nh), int(nw))
        assert got.shape == (nh, nw, 3)
        np.testing.assert_array_equal(got, g9[f"out{i}"])


def test_resize_shortest_edge_shapes():
    assert R.resize_shortest_edge_shape(480, 640, 800, 1333) == (800, 1067)
    assert R.resize_shortest_edge_shape(400, 1000, 800, 1333) == (533, 1333)
    assert R.resize_shortest_edge_shape(90, 130, 96, 160) == (96, 139)
