"""CPU gate on the kernels' register budget, read from the code objects INSIDE the shipped librls_mi355x.so
(tools/kernel_metadata.py: clang offload bundles -> AMDGPU metadata notes; no compiler and no GPU needed):

* every kernel a BASELINE config launches uses NO scratch (a spill inside an iteration loop is a memory round trip per
  iteration on kernels that are tuned to the microsecond);
* the kernels that do spill are exactly the known ones below, each with the shape that reaches it -- a new spill anywhere
  else fails here instead of showing up as an unexplained slowdown on the GPU box."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels():
    import kernel_metadata

    if not os.path.exists(kernel_metadata.LIB):
        pytest.skip("library not built")
    ks = kernel_metadata.kernels()
    assert len(ks) > 300, "the metadata parser lost the kernels"
    return ks


# (config, regular expression on the demangled instantiation) -- what each BASELINE config launches on its default path
BASELINE = [
    ("headline / configs[3] single solves: resident CGNR 4096x2048 CF32", r"cgnr_resident_kernel<c32, 8, 32, 8, 2, true, false>"),
    ("headline shape, solve! loop with callbacks: the listening kernel that runs one iteration ahead (SPEC)", r"cgnr_resident_kernel<c32, 8, 32, 8, 2, true, true>"),
    ("headline on the two-launch pipeline", r"cgnr_pipe_a_kernel<c32, 8, 32, 8, true, false, (true|false), false>"),
    ("headline on the two-launch pipeline", r"cgnr_pipe_r_kernel<c32>"),
    ("headline on the two-launch pipeline", r"cgnr_pipe_f_kernel<c32, \d+>"),
    ("configs[0]: CGNR 256x128 F32, single-workgroup kernel", r"cgnr_small_kernel<float, 8, 8>"),
    ("configs[0] on the pipeline (small = 0)", r"cgnr_pipe_a_kernel<float, 8, 8, 8, (true|false), false, (true|false), false>"),
    ("configs[1]: FISTA + L1 4096x2048 CF32, resident (last argument: the listening kernel that runs one iteration ahead)", r"fista_resident_kernel<c32, 8, 32, 8, 2, true, (true|false)>"),
    ("headline / configs[1] shape on the reference's default operator (AHA explicit), resident; last argument 1 / 2: listens / and runs ahead", r"(cgnr|fista)_gram_resident_kernel<c32, 16, 1, true, (0|1|2)>"),
    ("configs[1] on the pipeline", r"fista_pipe_a_kernel<c32, 8, 32, 8, true, (true|false), false>"),
    ("configs[1] shape, SURVEY 8f-1: OptISTA / POGM (2: with gradient restart) blocks of iterations as resident launches", r"pgm_resident_kernel<c32, 8, 32, 8, 2, true, (0|1|2)>"),
    ("configs[1] on the pipeline", r"fista_pipe_r_kernel<c32>"),
    ("configs[2]: ADMM + TV 8192x4096 F32: cg! on the pipeline, 512 row blocks walked by 256 workgroups (last argument)", r"cgnr_pipe_a_kernel<float, 4, 32, 8, true, false, (true|false), true>"),
    ("Gram mode, ComplexF32 N in (2048, 4096]: 512 row blocks of AHA walked", r"(cgnr|fista)_gram_kernel<c32, 4, 32, 8, (true|false), ((true|false), )?true>"),
    ("shapes with more row blocks than CUs: plain normal operator, FISTA", r"normal_slab_multi_kernel<(float|c32), \d, 32, 8, (true|false)>"),
    ("shapes with more row blocks than CUs: plain normal operator, FISTA", r"fista_pipe_a_kernel<(float|c32), \d, 32, 8, (true|false), (true|false), true>"),
    ("configs[2]: cg! entry, z / u update", r"cg_pipe_start_kernel<float>"),
    ("configs[2]: cg! entry, z / u update", r"admm_zu_kernel<float>"),
    ("configs[2]: TV prox of a 64 x 64 image", r"fgp2d_kernel<float, 4>"),
    ("configs[3]: 8 right-hand sides per GPU on the matrix cores", r"skinny_t_kernel<c32, 4, 4, true, 8, false>"),
    ("configs[3] on the reference's default operator (AHA explicit): one product per iteration", r"skinny_t_kernel<c32, 4, 4, true, 8, true>"),
    ("configs[3], AHA explicit, register-resident", r"cgnr_gramk_resident_kernel<\d, (true|false)>"),
    ("configs[3], AHA explicit, register-resident, FISTA columns", r"fista_gramk_resident_kernel<\d, (true|false)>"),
    ("configs[3]", r"skinny_v_kernel<c32, 4, 1, true, 2>"),
    ("configs[3]", r"skinny_u_kernel<c32, false, 4, 512>"),
    ("configs[4]: row shards 8192x8192 CF32, two GEMVs + update", r"gemv_n_kernel<c32.*>"),
    ("configs[4]", r"gemv_t_kernel<c32.*>"),
    ("configs[4]", r"cgnr_update_(reg_)?kernel<c32.*>"),
    ("configs[4]: the direct all-reduce", r"comm_(push|sum)_kernel"),
]

# kernels that are allowed to spill, with the shape that reaches them.  Empty since round 4: the three entries of round 3 went
# (fgp2d_kernel<float, 8> forms its neighbour indices from the masks and reads xTmp back from LDS, fgp2d_kernel<c32, 8> was
# never launched and is no longer instantiated; cgnr_gram_kernel<c32, 4, 32, 8, *> reads x where workgroup 0 updates it;
# cgnr_pipe_a_kernel<c32, 4, 32, 8, *, false, false> is not instantiated -- an unhinted launch runs the hinted kernel on a guess).
KNOWN_SPILLS = {}


def test_baseline_kernels_use_no_scratch(kernels):
    bad = []
    for cfg, pat in BASELINE:
        hit = [k for k in kernels if re.fullmatch(pat, k)]
        assert hit, f"{cfg}: no kernel matches {pat} (renamed? update this table)"
        bad += [f"{cfg}: {k} spills {kernels[k]['scratch']} B/lane" for k in hit if kernels[k]["scratch"] > 0]
    assert not bad, "\n".join(bad)


def test_spills_are_only_the_known_ones(kernels):
    """no kernel of the shipped library uses scratch memory (KNOWN_SPILLS is the place to argue an exception)"""
    spill = {k: v["scratch"] for k, v in kernels.items() if v["scratch"] > 0}
    unknown = [f"{k}: {b} B/lane" for k, b in spill.items() if not any(re.fullmatch(p, k) for p in KNOWN_SPILLS)]
    assert not unknown, "new spilling kernels:\n" + "\n".join(unknown)
    for p, why in KNOWN_SPILLS.items():   # the table must not rot either: an entry nothing matches any more goes
        assert any(re.fullmatch(p, k) for k in spill), f"KNOWN_SPILLS entry no longer spills (remove it): {p} ({why})"
    assert all(b <= 400 for b in spill.values())
    for cfg, pat in BASELINE:
        assert not any(re.fullmatch(p, k) for k in kernels if re.fullmatch(pat, k) for p in KNOWN_SPILLS), cfg


def test_register_budgets_admit_the_intended_occupancy(kernels):
    """the resident kernels need ONE 512-thread workgroup per CU: at most 256 VGPRs; the batched matrix-core kernels run one
    wave per SIMD by design (skinny.hip) and must stay under 512"""
    for k, v in kernels.items():
        if "resident_kernel" in k:
            assert v["vgprs"] <= 256, (k, v)
        assert v["vgprs"] <= 512, (k, v)


def _vregs(tok):
    """registers named by an operand token: 'v7' -> {7}, 'v[24:25]' -> {24, 25}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def test_hand_issued_loads_are_not_touched_before_their_hand_written_wait():
    """`cgnr_gramk_resident_kernel` (gramk.hip) issues its 16 partial-dot loads as inline-asm `buffer_load_dwordx2 ... sc1` and waits
    for them with a hand-written `s_waitcnt vmcnt(2 NE)` -- the compiler does not know those registers are in flight, so nothing in
    the SHIPPED code may read, copy, spill or overwrite them between the loads and that wait, the V loads behind them must be
    exactly the 2 NE the count assumes, and the first use must come after the wait (the advisor's round-4 finding; this is the
    ISA-level check it asked for, on the code objects inside librls_mi355x.so)."""
    import kernel_metadata

    if not os.path.exists(kernel_metadata.LIB):
        pytest.skip("library not built")
    ks = kernel_metadata.disassemble("cgnr_gramk_resident_kernel")
    assert len(ks) >= 4, sorted(ks)
    for name, ins in ks.items():
        ne = int(re.search(r"<(\d+),", name).group(1))
        loads = [i for i, l in enumerate(ins) if l.startswith("buffer_load_dwordx2") and " sc1" in l]
        assert len(loads) == 16, (name, len(loads))
        dst = set()
        for i in loads:
            dst |= _vregs(ins[i].split()[1].rstrip(","))
        assert len(dst) == 32, (name, "the 16 destinations overlap")
        # the wait: the first s_waitcnt with a vmcnt behind the last hand-issued load
        w = next(i for i in range(loads[-1] + 1, len(ins)) if ins[i].startswith("s_waitcnt") and "vmcnt" in ins[i])
        assert re.search(r"vmcnt\((\d+)\)", ins[w]).group(1) == str(2 * ne), (name, ins[w])
        between = ins[loads[0] + 1:w]
        later_vmem = [l for l in ins[loads[-1] + 1:w] if l.startswith(("buffer_load", "global_load", "flat_load", "scratch_load"))]
        assert len(later_vmem) == 2 * ne and all(l.startswith("buffer_load_dwordx4") for l in later_vmem), (name, later_vmem)
        for l in between:
            if l.startswith("buffer_load_dwordx2") and " sc1" in l:
                continue
            assert not l.startswith(("scratch_", "s_waitcnt")), (name, l)
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", l)
            hit = set().union(*[_vregs(t) for t in toks]) & dst if toks else set()
            # the V loads may not land in, nor address through, a register that is still in flight
            assert not hit, (name, l, sorted(hit))
        # and the sum is formed only behind the wait
        assert any(l.startswith("v_add_f64") for l in ins[w + 1:w + 8]), (name, ins[w + 1:w + 8])
