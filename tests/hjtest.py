"""Shared helpers for the test-suite."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pkg():
    return graft.load_package()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def sorted_triples(k, pr, ps):
    ku, pru, psu = (np.asarray(x, np.int32).view(np.uint32) for x in (k, pr, ps))
    o = np.lexsort((psu, pru, ku))
    return ku[o], pru[o], psu[o]
