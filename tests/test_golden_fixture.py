"""The committed golden fixture must still be what the oracle produces (guards the
oracle against accidental change; CPU-only)."""
import os

import numpy as np

from tests import _oracle as O


def test_flat_10kx128_fixture_matches_oracle():
    g = np.load(os.path.join(O.ROOT, "tests", "golden", "flat_10kx128_cosine.npz"))
    rows = O.gen_rows(int(g["corpus_seed"]), 0, 10000, 128)
    qs = O.gen_rows(int(g["query_seed"]), 0, g["rows"].shape[0], 128)
    for i in range(0, g["rows"].shape[0], 5):
        r, d = O.exact_search(0, rows, qs[i], 10)
        assert np.array_equal(r, g["rows"][i])
        assert np.array_equal(d.view(np.uint32), g["dist"][i].view(np.uint32))
