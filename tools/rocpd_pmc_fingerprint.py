"""Fingerprint of the kernel sources a PMC summary was measured on (shared by tools/rocpd_pmc.py and bench.py)."""
import hashlib
import os


def csrc_fingerprint():
    """sha256 over the kernel sources (csrc/*.hip, *.h, Makefile) in name order."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sylph-few-shot-detection_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(root, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]
