"""The arithmetic behind the latency form of the HNSW traversal (quiver_amd/csrc/qv_hnsw.hip, "a row's sum over several lanes,
certified"): the reference accumulates a distance as ONE float64 chain in element order (pkg/vectortypes/distances.go:18-22); the
latency form adds the same exact products as many partial chains and returns the float32 only when the interval
[S - B, S + B] around its sum maps to a single float32 through the (monotone) finalisation.  Checked here on the CPU, in
numpy float64 (IEEE, same roundings as the device's v_fma_f64 on exact products):
  * the reference's chain always lies inside [S - B, S + B] for the bound the kernel uses;
  * the finalisation is monotone in the sum, so equal ends decide the float32;
  * the bound is small enough that the certificate holds for all but a tiny share of evaluations."""
import numpy as np
import pytest

U = 2.0 ** -53
SLACK = 128.0


def seq_sum(p):                       # one chain, element order: cumsum adds left to right, one rounding per step
    return float(np.cumsum(p)[-1]) if len(p) else 0.0


def split_sum(p, waves, lanes_per_row):
    """the latency form's order: wave w owns a contiguous run of 16-byte chunks (4 elements each), its lanes split the run,
    every lane runs a chain, lanes are added pairwise (butterfly), waves in order"""
    n4 = len(p) // 4
    n_p = n4 // 8                                        # pieces of 8 chunks
    tot = None
    for w in range(waves):
        p_lo, p_hi = w * n_p // waves, (w + 1) * n_p // waves
        c0, n = p_lo * 8, (p_hi - p_lo) * 8
        parts = []
        for sub in range(lanes_per_row):
            lo, hi = c0 + sub * n // lanes_per_row, c0 + (sub + 1) * n // lanes_per_row
            parts.append(seq_sum(p[4 * lo:4 * hi]))
        while len(parts) > 1:                             # xor-butterfly: (a + b), then pairs of pairs
            half = len(parts) // 2
            parts = [parts[i] + parts[i + half] for i in range(half)]
        tot = parts[0] if tot is None else tot + parts[0]
    return tot


def cosine_finalize(acc, qn, rn):     # distances.go:25-39 (float64, then float32)
    if qn == 0.0 or rn == 0.0:
        return np.float32(1.0)
    sim = acc / (qn * rn)
    sim = min(1.0, max(-1.0, sim))
    return np.float32(1.0 - sim)


@pytest.mark.parametrize("dim", [32, 96, 128, 768, 1024])
@pytest.mark.parametrize("lanes_per_row", [2, 4, 8])
def test_reference_chain_lies_in_the_certified_interval(dim, lanes_per_row):
    rng = np.random.default_rng(dim * 10 + lanes_per_row)
    undecided = 0
    trials = 400
    for t in range(trials):
        scale = 10.0 ** rng.integers(-3, 4)
        q = (rng.standard_normal(dim) * scale).astype(np.float32)
        r = (rng.standard_normal(dim) * scale).astype(np.float32)
        if t % 5 == 0:
            r = (q * np.float32(1.0 + 1e-3 * rng.standard_normal())).astype(np.float32)      # nearly parallel: the clamp's neighbourhood
        p = q.astype(np.float64) * r.astype(np.float64)                  # exact: 24 x 24 bits
        s_ref = seq_sum(p)
        s = split_sum(p, 8, lanes_per_row)
        qn = float(np.sqrt(seq_sum(q.astype(np.float64) ** 2))); rn = float(np.sqrt(seq_sum(r.astype(np.float64) ** 2)))
        b = (2.0 * dim + SLACK) * U * qn * rn                            # split_bound<QV_COSINE>
        assert s - b <= s_ref <= s + b, (dim, t, s, s_ref, b)
        # and with room to spare: the a-priori bound, g(h) sum|p|, is itself below b
        assert abs(s - s_ref) <= (dim + dim // lanes_per_row + 16) * U * float(np.sum(np.abs(p))) * 1.0001 + 1e-300
        d_lo, d_hi, d_ref = cosine_finalize(s - b, qn, rn), cosine_finalize(s + b, qn, rn), cosine_finalize(s_ref, qn, rn)
        assert d_hi <= d_ref <= d_lo                                     # monotone (non-increasing in the sum)
        if d_lo.tobytes() == d_hi.tobytes():
            assert d_ref.tobytes() == d_lo.tobytes()                     # the certificate decides the float32
        else:
            undecided += 1                                               # the kernel walks the row again as one chain
    assert undecided <= trials // 4                                      # (only the nearly-parallel pairs: distances near 0)


def test_nonnegative_terms_bound_is_the_sum_itself():
    """L2 / L1 / squared-L2-in-float64: every term is >= 0, so sum|t_i| is the sum and B = k_u * S"""
    rng = np.random.default_rng(7)
    for dim in (32, 160, 768):
        for _ in range(200):
            a = rng.standard_normal(dim).astype(np.float32); b_ = rng.standard_normal(dim).astype(np.float32)
            d = (a - b_).astype(np.float64)                              # float32 subtract, widened (distances.go:50)
            t = d * d                                                    # exact
            s_ref, s = seq_sum(t), split_sum(t, 8, 4)
            b = (2.0 * dim + SLACK) * U * s
            assert s - b <= s_ref <= s + b
            lo, hi, ref = np.float32(np.sqrt(max(s - b, 0.0))), np.float32(np.sqrt(s + b)), np.float32(np.sqrt(s_ref))
            assert lo <= ref <= hi
