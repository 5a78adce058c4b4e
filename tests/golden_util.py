import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))


def load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def rows_text(rows, idx):
    """(read string, ref string) of one alignment, as the reference host prints them."""
    s = int(idx[0])
    return bytes(rows[0, s:-1]), bytes(rows[1, s:-1])
