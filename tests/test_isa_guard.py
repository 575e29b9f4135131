"""Static guard on the compiled flat-scan loops (CPU-only: hipcc cross-compiles).

The scan is HBM-bound only if each wave keeps its 16 row-chunk loads in flight together; hipcc is free to sink loads next
to their uses, and did for every metric but cosine (2 loads in flight: 52-78 % of the HBM peak instead of 88-90 %) until
a scheduling barrier pinned them.  This test compiles qv_scan.hip to assembly and requires, for every metric's
k_flat_scan<M,16>, a point in the instruction stream where >= 16 global_load_dwordx4 are outstanding (counting each
load and clipping at every s_waitcnt vmcnt(N))."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_every_metric_keeps_16_loads_in_flight(tmp_path):
    asm = str(tmp_path / "scan.s")
    src = os.path.join(ROOT, "quiver_amd", "csrc", "qv_scan.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", asm, src],
                   check=True, capture_output=True, text=True, cwd=os.path.dirname(src))
    text = open(asm).read()
    for metric, fused in [(m_, f_) for m_ in range(9) for f_ in (0, 1)]:      # fused: the single-launch form (the last workgroup merges), round 5
        m = re.search(r"^_ZN2qv11k_flat_scanILi%dELi16ELb%dEEEv\w*:[^\n]*\n(.*?)\n\s+s_endpgm" % (metric, fused), text, re.S | re.M)
        assert m, "k_flat_scan<%d,16,%d> not found" % (metric, fused)
        # outstanding vector loads along the instruction stream: +1 per row-chunk load, clipped by every s_waitcnt vmcnt(N)
        best = out = 0
        for line in m.group(1).split("\n"):
            t = line.strip()
            if t.startswith("global_load_dwordx4"):
                out += 1; best = max(best, out)
            elif t.startswith("s_waitcnt"):
                w = re.search(r"vmcnt\((\d+)\)", t)
                if w:
                    out = min(out, int(w.group(1)))
            elif t.startswith(".LBB") or t.startswith("s_cbranch") or t.startswith("s_branch"):
                pass                                            # straight-line approximation: loop bodies are what matters
        need = 8 if metric == 0 else 16                         # cosine: hipcc's own rolling window (measured best); others: the pinned batch
        assert best >= need, "metric %d: at most %d row-chunk loads in flight" % (metric, best)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_key_writing_scan_keeps_the_flat_scan_loop(tmp_path):
    """k_flat_keys (one key per row, for k above 128 and the full ranking) walks its tiles with the flat scan's loop: measured 4.40
    against 4.47 ms at 10M x 768 when its requests are pinned per block instead (round 4).  Same count as above on its compiled loop."""
    asm = str(tmp_path / "rank.s")
    src = os.path.join(ROOT, "quiver_amd", "csrc", "qv_rank.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", asm, src],
                   check=True, capture_output=True, text=True, cwd=os.path.dirname(src))
    text = open(asm).read()
    for metric in range(9):
        for hist in (0, 1):
            m = re.search(r"^_ZN2qv11k_flat_keysILi%dELi16ELb%dEEEv\w*:[^\n]*\n(.*?)\n\s+s_endpgm" % (metric, hist), text, re.S | re.M)
            assert m, "k_flat_keys<%d,16,%d> not found" % (metric, hist)
            best = out = 0
            for line in m.group(1).split("\n"):
                t = line.strip()
                if t.startswith("global_load_dwordx4"):
                    out += 1; best = max(best, out)
                elif t.startswith("s_waitcnt"):
                    w = re.search(r"vmcnt\((\d+)\)", t)
                    if w:
                        out = min(out, int(w.group(1)))
            need = 8 if metric == 0 else 16
            assert best >= need, "k_flat_keys<%d,16,%d>: at most %d row-chunk loads in flight" % (metric, hist, best)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_query_resident_filter_loop_is_what_was_written(tmp_path):
    """k_qreg_filter (qv_qreg.hip) keeps 64 queries' operands in registers and streams rows by LDS-DMA issued from inline assembly, which
    the compiler cannot see.  Three things make or break it, all visible in the compiled tile loop (between a tile's first and last
    matrix instruction): the A operands are read where they live (no v_accvgpr_read copies), nothing spills into the loop, and
    the only vector-memory waits are the counted ones written by hand — a compiler-inserted s_waitcnt vmcnt(0) (e.g. for an operand's
    load at its first use) would drain the whole row ring once per tile (measured: 482 against 391 us)."""
    asm = str(tmp_path / "qreg.s")
    src = os.path.join(ROOT, "quiver_amd", "csrc", "qv_qreg.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", asm, src],
                   check=True, capture_output=True, text=True, cwd=os.path.dirname(src))
    text = open(asm).read()
    kernels = re.findall(r"^(_ZN2qv13k_qreg_filterI\w+):[^\n]*\n(.*?)\n\s+s_endpgm", text, re.S | re.M)
    assert len(kernels) >= 18, "expected 3 metrics x 3 widths x 2 row formats, found %d" % len(kernels)
    for name, body in kernels:
        lines = [l.strip() for l in body.split("\n")]
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma_f32_32x32x16_bf16")]
        steps = int(re.search(r"k_qreg_filterILi\dELi(\d+)E", name).group(1))
        assert len(mf) == 4 * steps, "%s: %d matrix instructions for %d steps" % (name, len(mf), steps)
        loop = lines[mf[0]: mf[-1] + 1]
        assert not any(l.startswith("v_accvgpr") for l in loop), name + ": operand copies in the tile loop"
        assert not any(l.startswith("scratch_") for l in loop), name + ": spill traffic in the tile loop"
        waits = [int(w) for l in loop for w in re.findall(r"vmcnt\((\d+)\)", l)]
        assert waits and min(waits) >= 8, "%s: a vector-memory wait of %d in the tile loop drains the row ring" % (name, min(waits) if waits else -1)
        assert sum(1 for l in loop if l.startswith("global_load_lds_dwordx4")) >= 4, name + ": no row requests in the tile loop"
        # The matrix instructions are inline assembly: the compiler's hazard recogniser does not look inside.  A vector-ALU write of a
        # register (a conversion, a copy) needs two wait states before a matrix instruction reads it as A or B.
        def regs(tok):
            m2 = re.match(r"([va])\[(\d+):(\d+)\]", tok) or re.match(r"([va])(\d+)$", tok)
            if not m2:
                return set()
            lo = int(m2.group(2)); hi = int(m2.group(3)) if m2.lastindex == 3 else lo
            return {(m2.group(1), r) for r in range(lo, hi + 1)}
        recent = []                                             # (registers written, wait states since) of the last vector-ALU writes
        for l in loop:
            if not l or l.startswith(";") or l.startswith("."):
                continue
            ops = [t.strip() for t in l.split(None, 1)[1].split(",")] if " " in l else []
            if l.startswith("v_mfma"):
                src = regs(ops[1]) | regs(ops[2])
                for written, ws in recent:
                    assert not (written & src) or ws >= 2, "%s: %s reads a register a vector instruction wrote %d wait states before" % (name, l, ws)
            gain = int(l.split()[1]) + 1 if l.startswith("s_nop") else 1
            recent = [(w, ws + gain) for w, ws in recent if ws + gain < 8]
            if l.startswith("v_") and not l.startswith("v_mfma") and not l.startswith("v_cmp") and ops:
                recent.append((regs(ops[0]), 0))
