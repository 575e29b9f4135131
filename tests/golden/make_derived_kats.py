"""Writes tests/golden/derived_kats.json: hand-derivable known-answer vectors that DISCRIMINATE between the reference's
arithmetic (pkg/vectortypes/distances.go:12-104, pkg/hnsw/adapter.go:105-167, pkg/index/arrow_hnsw.go:124-132) and the
plausible wrong restatements of it — float32 instead of float64 accumulation, "widen then subtract" instead of "subtract in
float32, then widen", fused instead of separately rounded multiply-add, reversed or pairwise summation, a missing clamp.
The reference's own tests (tests/golden/ref_kats.json) hold 27 distance vectors of dimension <= 3 at tolerance 1e-6: none of
them can tell these apart.

Every vector carries the expected float32 BITS, the derivation in words, and what each wrong restatement would return (which
must differ from the expected bits, or the vector discriminates nothing: checked below).  The expected bits are computed here
by a THIRD statement of the reference's expressions, independent of oracle/qv_oracle.c and oracle/oracle_np.py: float64
operations are Python float operations (IEEE double, never fused), float32 operations are exact rationals rounded once to 24
bits (round-half-even) — no numpy, no C.  The derivation strings were written by hand; this script fails if the bits stated
in a derivation disagree with what it computes.

    python tests/golden/make_derived_kats.py        # rewrites derived_kats.json next to it
"""
import json
import math
import os
import struct
from fractions import Fraction as Fr


def rn32(x) -> float:
    """exact rational (or float) -> nearest float32 (ties to even), returned as a Python float holding that value"""
    x = Fr(x)
    if x == 0:
        return 0.0
    s = -1 if x < 0 else 1
    x = abs(x)
    e = math.floor(math.log2(float(x))) if x >= Fr(1, 2 ** 1000) else -1000
    while Fr(2) ** e > x:
        e -= 1
    while Fr(2) ** (e + 1) <= x:
        e += 1
    e = max(e, -126)                                   # subnormals share the exponent of the smallest normal
    q = x / Fr(2) ** (e - 23)                          # in units of the last place
    n = q.numerator // q.denominator
    r = q - n
    if r > Fr(1, 2) or (r == Fr(1, 2) and n % 2 == 1):
        n += 1
    v = Fr(n) * Fr(2) ** (e - 23)
    assert v < Fr(2) ** 128
    return float(s * v)                                # a float32 value is exactly a float64 value


def bits(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", x))[0]


def f32(x):                                            # the inputs are float32 values
    assert rn32(x) == float(x), x
    return float(x)


# ---- the reference's expressions.  f64: Python float ops.  f32: rn32 of the exact rational result of each operation. ----
def add32(a, b): return rn32(Fr(a) + Fr(b))
def sub32(a, b): return rn32(Fr(a) - Fr(b))
def mul32(a, b): return rn32(Fr(a) * Fr(b))
def div32(a, b): return rn32(Fr(a) / Fr(b))


def cosine(a, b, clamp=True, sqrt_of_product=False, f32acc=False):          # distances.go:12-40
    dot = ma = mb = 0.0
    for x, y in zip(a, b):
        if f32acc:
            dot, ma, mb = add32(dot, mul32(x, y)), add32(ma, mul32(x, x)), add32(mb, mul32(y, y))
        else:
            dot += x * y; ma += x * x; mb += y * y                          # :18-22 (products of float32 values are exact in float64)
    if ma == 0 or mb == 0:
        return 1.0                                                         # :25-27
    sim = dot / (math.sqrt(ma * mb) if sqrt_of_product else math.sqrt(ma) * math.sqrt(mb))   # :30
    if clamp:
        sim = max(-1.0, min(1.0, sim))                                     # :32-36
    return rn32(1.0 - sim)                                                 # :39


def euclidean(a, b, widen_first=False):                                    # distances.go:43-55
    s = 0.0
    for x, y in zip(a, b):
        d = (x - y) if widen_first else sub32(x, y)                        # :50 float64(a[i] - b[i]): the subtraction is float32
        s += d * d
    return rn32(math.sqrt(s))


def sqeuclidean(a, b, fused=False, f64acc=False, reverse=False):           # distances.go:60-72: all float32
    s = 0.0
    pairs = list(zip(a, b))
    if reverse:
        pairs.reverse()
    for x, y in pairs:
        d = sub32(x, y)
        if f64acc:
            s += d * d
        elif fused:
            s = rn32(Fr(s) + Fr(d) * Fr(d))
        else:
            s = add32(s, mul32(d, d))
    return rn32(s)


def dotdist(a, b, f32acc=False):                                           # distances.go:77-90
    dot = 0.0
    for x, y in zip(a, b):
        dot = add32(dot, mul32(x, y)) if f32acc else dot + x * y
    return sub32(1.0, dot) if f32acc else rn32(1.0 - dot)


def manhattan(a, b, widen_first=False):                                    # distances.go:93-104
    s = 0.0
    for x, y in zip(a, b):
        s += abs((x - y) if widen_first else sub32(x, y))
    return rn32(s)


def cosine_f32(a, b, f64acc=False):                                        # adapter.go:105-136
    if f64acc:
        return cosine(a, b)
    dot = na = nb = 0.0
    for x, y in zip(a, b):
        dot, na, nb = add32(dot, mul32(x, y)), add32(na, mul32(x, x)), add32(nb, mul32(y, y))
    if na == 0 or nb == 0:
        return 1.0
    den = mul32(rn32(math.sqrt(na)), rn32(math.sqrt(nb)))                  # :128
    sim = div32(dot, den)
    sim = max(-1.0, min(1.0, sim))
    return sub32(1.0, sim)


def euclidean_f32(a, b, f64acc=False):                                     # adapter.go:139-151
    s = 0.0
    for x, y in zip(a, b):
        d = sub32(x, y)
        s = s + d * d if f64acc else add32(s, mul32(d, d))
    return rn32(math.sqrt(s))


def dot_f32(a, b, f64acc=False):                                           # adapter.go:154-165
    dot = 0.0
    for x, y in zip(a, b):
        dot = dot + x * y if f64acc else add32(dot, mul32(x, y))
    return rn32(1.0 - dot) if f64acc else sub32(1.0, dot)


def sqeuclidean_f64(a, b, subtract_f32=False):                             # index/arrow_hnsw.go:124-132
    s = 0.0
    for x, y in zip(a, b):
        d = sub32(x, y) if subtract_f32 else x - y
        s += d * d
    return rn32(s)


P = lambda e: 2.0 ** e

KATS = [
    dict(name="dot_needs_float64_accumulation", metric=3, cites="pkg/vectortypes/distances.go:83-89",
         a=[1.0, P(-20)], b=[1.0, P(-20)], want_bits=0xAB800000,
         derivation="dot = 1*1 + 2^-20*2^-20 = 1 + 2^-40, which float64 holds exactly (41 significant bits); 1.0 - dot = -2^-40, a "
                    "power of two, so float32(-2^-40) is exact: sign 1, exponent 127-40 = 87 = 0x57, mantissa 0 -> 0xAB800000.  "
                    "A float32 accumulator rounds 1 + 2^-40 to 1 and returns 0 = 0x00000000.",
         fn=lambda a, b: dotdist(a, b), wrong={"float32_accumulator": lambda a, b: dotdist(a, b, f32acc=True)}),
    dict(name="cosine_needs_the_clamp", metric=0, cites="pkg/vectortypes/distances.go:30-39",
         a=[1.0, 1.0, 1.0], b=[1.0, 1.0, 1.0], want_bits=0x00000000,
         derivation="dot = |a|^2 = |b|^2 = 3.  sqrt(3) rounds DOWN to 1.7320508075688772, and 1.7320508075688772^2 rounds to "
                    "2.9999999999999996 = 3 - 2^-51, so similarity = 3 / (3 - 2^-51) = 1 + 2^-52 > 1: the clamp (:32-36) makes it 1 and "
                    "the distance float32(1.0 - 1.0) = +0 -> 0x00000000.  Without the clamp: float32(-2^-52) = 0xA5800000.",
         fn=lambda a, b: cosine(a, b), wrong={"no_clamp": lambda a, b: cosine(a, b, clamp=False)}),
    dict(name="cosine_multiplies_two_square_roots", metric=0, cites="pkg/vectortypes/distances.go:30",
         a=[1.0, 1.0], b=[1.0, 1.0], want_bits=0x25800000,
         derivation="dot = |a|^2 = |b|^2 = 2.  sqrt(2) rounds UP to 1.4142135623730951 and its square rounds to 2.0000000000000004 = "
                    "2 + 2^-51, so similarity = 2 / (2 + 2^-51) = 1/(1 + 2^-52) -> 1 - 2^-52 (the neighbour of 1 below is 1 - 2^-53; "
                    "1 - 2^-52 + 2^-104 rounds to 1 - 2^-52), and the distance of two IDENTICAL vectors is float32(2^-52): exponent "
                    "127-52 = 75 = 0x4B -> 0x25800000, not 0.  sqrt(|a|^2 * |b|^2) = sqrt(4) = 2 would give exactly 0 = 0x00000000.",
         fn=lambda a, b: cosine(a, b), wrong={"sqrt_of_the_product": lambda a, b: cosine(a, b, sqrt_of_product=True)}),
    dict(name="euclidean_subtracts_in_float32_before_widening", metric=1, cites="pkg/vectortypes/distances.go:49-52",
         a=[16777218.0] * 3, b=[0.75] * 3, want_bits=0x4BDDB3D9,
         derivation="a[i] - b[i] = 16777217.25 lies between the float32 neighbours 16777216 and 16777218 (spacing 2 above 2^24) and rounds "
                    "to 16777218; sum = 3 * 16777218^2 = 844425131458572 (exact in float64: < 2^53); sqrt = sqrt(3) * 16777218 = 29058993.98...; spacing 2 -> "
                    "float32 29058994 = 0x4BDDB3D9.  Widening first keeps 16777217.25: sqrt(3) * 16777217.25 = 29058992.68... -> 29058992 = "
                    "0x4BDDB3D8.",
         fn=lambda a, b: euclidean(a, b), wrong={"widen_then_subtract": lambda a, b: euclidean(a, b, widen_first=True)}),
    dict(name="manhattan_subtracts_in_float32_before_widening", metric=4, cites="pkg/vectortypes/distances.go:99-101",
         a=[16777218.0] * 3, b=[0.75] * 3, want_bits=0x4C400002,
         derivation="each float32 difference is 16777218 (as above); sum = 50331654, halfway between the float32 neighbours 50331652 and "
                    "50331656 (spacing 4 above 2^25): ties-to-even picks 50331656 = 12582914 * 4 = 0x4C400002.  Widening first sums "
                    "3 * 16777217.25 = 50331651.75 -> 50331652 = 0x4C400001.",
         fn=lambda a, b: manhattan(a, b), wrong={"widen_then_subtract": lambda a, b: manhattan(a, b, widen_first=True)}),
    dict(name="squared_euclidean_is_a_sequential_float32_sum", metric=2, cites="pkg/vectortypes/distances.go:65-71",
         a=[4096.0, 1.0, 1.0, 1.0, 1.0], b=[0.0] * 5, want_bits=0x4B800000,
         derivation="sum = 4096^2 = 2^24; then four times 2^24 + 1, which is halfway between 2^24 and 2^24 + 2 and rounds to the even "
                    "2^24: the ones are lost one by one, result 16777216 = 0x4B800000.  A float64 accumulator, the reverse order "
                    "(1+1+1+1 = 4, then 4 + 2^24) or a pairwise sum all give 16777220 = 0x4B800002.",
         fn=lambda a, b: sqeuclidean(a, b), wrong={"float64_accumulator": lambda a, b: sqeuclidean(a, b, f64acc=True),
                                                    "reverse_order": lambda a, b: sqeuclidean(a, b, reverse=True)}),
    dict(name="squared_euclidean_rounds_the_product_before_the_add", metric=2, cites="pkg/vectortypes/distances.go:69-70",
         a=[P(-12), 1.0 + P(-12)], b=[0.0, 0.0], want_bits=0x3F801000,
         derivation="sum = (2^-12)^2 = 2^-24.  diff^2 = (1 + 2^-12)^2 = 1 + 2^-11 + 2^-24 needs 25 bits: halfway between 1 + 2^-11 and "
                    "1 + 2^-11 + 2^-23, ties-to-even -> 1 + 2^-11.  Then 2^-24 + 1 + 2^-11 is the same tie again -> 1 + 2^-11 = "
                    "0x3F801000.  A fused multiply-add adds the exact square: 2^-24 + 1 + 2^-11 + 2^-24 = 1 + 2^-11 + 2^-23 = 0x3F801001 "
                    "(Go on amd64 does not fuse; the library is compiled with -ffp-contract=off for this).",
         fn=lambda a, b: sqeuclidean(a, b), wrong={"fused_multiply_add": lambda a, b: sqeuclidean(a, b, fused=True)}),
    dict(name="hnsw_cosine_accumulates_in_float32", metric=5, cites="pkg/hnsw/adapter.go:117-136",
         a=[4096.0, 1.0, 1.0, 1.0, 1.0], b=[4096.0, 0.0, 0.0, 0.0, 0.0], want_bits=0x00000000,
         derivation="float32 sums: dot = 2^24, normB = 2^24, normA = 2^24 + 1 + 1 + 1 + 1 with every + 1 lost to ties-to-even = 2^24; "
                    "sqrt = 4096 each, 4096 * 4096 = 2^24, similarity = 1, distance 1 - 1 = 0 -> 0x00000000 although the vectors differ.  "
                    "With float64 sums normA = 2^24 + 4, similarity = 1/sqrt(1 + 2^-22) = 1 - 2^-23 + 3 * 2^-47 - ..., distance "
                    "float32(2^-23 - 3 * 2^-47) = 0x33FFFFFD, three float32 steps below 2^-23 (vectortypes.CosineDistance on the same pair).",
         fn=lambda a, b: cosine_f32(a, b), wrong={"float64_accumulators": lambda a, b: cosine_f32(a, b, f64acc=True)}),
    dict(name="hnsw_euclidean_accumulates_in_float32", metric=6, cites="pkg/hnsw/adapter.go:144-150",
         a=[4096.0, 1.0, 1.0, 1.0, 1.0], b=[0.0] * 5, want_bits=0x45800000,
         derivation="float32 sum = 2^24 (the four ones are lost as above); sqrt(2^24) = 4096 = 0x45800000.  A float64 sum gives "
                    "sqrt(16777220) = 4096.000488..., spacing 2^-11 at 4096 -> 4096 + 2^-11 = 0x45800001.",
         fn=lambda a, b: euclidean_f32(a, b), wrong={"float64_accumulator": lambda a, b: euclidean_f32(a, b, f64acc=True)}),
    dict(name="hnsw_dot_accumulates_in_float32", metric=7, cites="pkg/hnsw/adapter.go:159-164",
         a=[4096.0, 1.0, 1.0, 1.0, 1.0], b=[4096.0, 1.0, 1.0, 1.0, 1.0], want_bits=0xCB7FFFFF,
         derivation="float32 dot = 2^24 (ones lost); 1 - 2^24 = -16777215, a 24-bit integer, exact: 0xCB7FFFFF.  A float64 dot is "
                    "16777220: 1 - 16777220 = -16777219, halfway between -16777218 and -16777220 -> even -16777220 = 0xCB800002.",
         fn=lambda a, b: dot_f32(a, b), wrong={"float64_accumulator": lambda a, b: dot_f32(a, b, f64acc=True)}),
    dict(name="arrow_squared_euclidean_widens_before_subtracting", metric=8, cites="pkg/index/arrow_hnsw.go:124-132",
         a=[16777218.0], b=[0.75], want_bits=0x57800001,
         derivation="both vectors are float64 there: d = 16777217.25 exactly, d^2 = 281475018653697.5625 (exact in float64: "
                    "67108869^2 / 16 < 2^53 / 16); float32 spacing at 2^48 is 2^25 = 33554432 and 281475018653697.5625 - 2^48 = "
                    "41943041.5625 lies above the midpoint 16777216 of the first step and below that of the second: 2^48 + 2^25 = "
                    "0x57800001.  Subtracting in float32 first gives 16777218^2 = 281475043819524 = 2^48 + 67108868, which is past the "
                    "midpoint 50331648 of the second step: 2^48 + 2^26 = 0x57800002.",
         fn=lambda a, b: sqeuclidean_f64(a, b), wrong={"subtract_in_float32": lambda a, b: sqeuclidean_f64(a, b, subtract_f32=True)}),
    dict(name="cosine_zero_vector_guard_sees_the_float64_norm", metric=0, cites="pkg/vectortypes/distances.go:25-27",
         a=[P(-80), 0.0], b=[1.0, 0.0], want_bits=0x00000000,
         derivation="|a|^2 = 2^-160 is far below float32's smallest subnormal (2^-149) but an ordinary float64: the zero guard does "
                    "not fire, similarity = 2^-80 / (2^-80 * 1) = 1, distance 0 -> 0x00000000.  With float32 norms |a|^2 underflows to 0 "
                    "and the guard returns 1 = 0x3F800000.",
         fn=lambda a, b: cosine(a, b), wrong={"float32_norms": lambda a, b: cosine(a, b, f32acc=True)}),
]


def main():
    out = []
    for k in KATS:
        a = [f32(x) for x in k["a"]]; b = [f32(x) for x in k["b"]]
        got = bits(k["fn"](a, b))
        assert got == k["want_bits"], (k["name"], hex(got), hex(k["want_bits"]))
        wrong = {}
        for label, fn in k["wrong"].items():
            w = bits(fn(a, b))
            assert w != got, (k["name"], label, "does not discriminate")
            assert ("0x%08X" % w) in k["derivation"], (k["name"], label, "0x%08X" % w, "not stated in the derivation")
            wrong[label] = "0x%08X" % w
        assert ("0x%08X" % got) in k["derivation"], (k["name"], "expected bits not stated in the derivation")
        out.append(dict(name=k["name"], metric=k["metric"], cites=k["cites"], a=a, b=b, want_bits="0x%08X" % got,
                        want=struct.unpack("<f", struct.pack("<I", got))[0], derivation=k["derivation"], wrong_restatements=wrong))
    doc = {"_comment": "Discriminating known-answer vectors derived by hand from the reference's source (see make_derived_kats.py, which "
                       "recomputes every value in exact rational / IEEE-double arithmetic and wrote this file).  Inputs are float32 values; "
                       "want_bits is the float32 result the reference's expression yields; wrong_restatements lists what a plausible "
                       "mis-restatement would return instead.", "distance": out}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "derived_kats.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", path, len(out), "vectors")


if __name__ == "__main__":
    main()
