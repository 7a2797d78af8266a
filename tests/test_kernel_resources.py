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
    ("headline / configs[3] single solves: resident CGNR 4096x2048 CF32", r"cgnr_resident_kernel<c32, 8, 32, 8, 2, true>"),
    ("headline on the two-launch pipeline", r"cgnr_pipe_a_kernel<c32, 8, 32, 8, true, false, (true|false)>"),
    ("headline on the two-launch pipeline", r"cgnr_pipe_r_kernel<c32>"),
    ("headline on the two-launch pipeline", r"cgnr_pipe_f_kernel<c32, \d+>"),
    ("configs[0]: CGNR 256x128 F32, single-workgroup kernel", r"cgnr_small_kernel<float, 8, 8>"),
    ("configs[0] on the pipeline (small = 0)", r"cgnr_pipe_a_kernel<float, 8, 8, 8, (true|false), false, (true|false)>"),
    ("configs[1]: FISTA + L1 4096x2048 CF32, resident", r"fista_resident_kernel<c32, 8, 32, 8, 2, true>"),
    ("configs[1] on the pipeline", r"fista_pipe_a_kernel<c32, 8, 32, 8, true, (true|false)>"),
    ("configs[1] shape, SURVEY 8f-1: OptISTA / POGM (2: with gradient restart) blocks of iterations as resident launches", r"pgm_resident_kernel<c32, 8, 32, 8, 2, true, (0|1|2)>"),
    ("configs[1] on the pipeline", r"fista_pipe_r_kernel<c32>"),
    ("configs[2]: ADMM + TV 8192x4096 F32: cg! on the pipeline", r"cgnr_pipe_a_kernel<float, 4, 32, 8, true, false, (true|false)>"),
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
