"""Contracts between what the hand-off protocols assume and what hipcc actually emits (CPU test: cross-compiles csrc/*.hip to gfx950 assembly,
no GPU needed).

dec_attn_pair_kernel (round 3): a workgroup signals its row behind `s_waitcnt vmcnt(0)` + the workgroup barrier, with NO vector-memory operation
between its four agent-scope partial stores and that wait -- in particular not the ten loads of the cross-attention half's projection operands,
which round 2 issued first (and counted: vmcnt(10)): a load instruction waits for room in the CU's vector-memory queue, which the other
workgroups' K/V streams keep full, and that wait sat in front of the signal (profiles/r03_notes.md: 245.5 -> 241.6 ms per batch).  If a
compiler change hoists those loads above the signal again, this test fails first.
dec_step_kernel (the option YMT3_STEP_KERNEL=1) keeps round 2's protocol: `s_waitcnt vmcnt(10)` behind exactly ten loads."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    return None


@pytest.fixture(scope="module")
def decode_asm(tmp_path_factory):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    from yourmt3_amd import build as B
    out = tmp_path_factory.mktemp("isa") / "decode.s"
    flags = [f for f in B.FLAGS if f not in ("-fPIC",)]
    cmd = [hipcc] + flags + ["-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "yourmt3_amd", "csrc", "decode.hip"),
                             "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


def _kernel_body(asm: str, name: str) -> list:
    m = re.search(r"^(_Z\w*" + name + r"\w*):[^\n]*\n(.*?)\n\s*s_endpgm", asm, re.S | re.M)
    assert m, f"kernel {name} not found in the assembly"
    return [l.strip() for l in m.group(2).splitlines() if l.strip() and not l.strip().startswith(";")]


VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)")


def test_attention_pair_signals_before_it_requests_the_projection_operands(decode_asm):
    body = _kernel_body(decode_asm, "dec_attn_pair_kernel")
    stores = [i for i, l in enumerate(body) if l.startswith("buffer_store_dwordx4") and " sc1" in l]
    assert len(stores) == 4, "the O-projection partial leaves as four 16-byte agent-scope stores per lane"
    atomics = [i for i, l in enumerate(body) if l.startswith("global_atomic_add") and i > stores[-1]]
    assert atomics, "the row's arrival"
    between = body[stores[-1] + 1:atomics[0]]
    # (a measurement-only stamp store sits behind a branch on the stamp pointer: older than the wait, covered by it)
    vmem = [l for l in between if VMEM.match(l) and not l.startswith("global_store_dwordx2")]
    assert not vmem, vmem
    waits = [l for l in between if l.startswith("s_waitcnt vmcnt(0)")]
    assert waits, "the stores' acknowledgement: s_waitcnt vmcnt(0)"
    assert sum(l.startswith("s_barrier") for l in between) == 1
    # the projection operands follow the arrival: 8 x wq, the residual element, the gain element
    after = [l for l in body[atomics[0] + 1:atomics[0] + 60] if VMEM.match(l)]
    assert len([l for l in after[:12] if l.startswith("global_load")]) >= 10, after[:12]


def test_attention_pair_stays_within_two_workgroups_per_cu(decode_asm):
    """512 workgroups of 512 threads must all be resident on 256 CUs: <= 128 VGPRs (4 waves per SIMD), no scratch."""
    m = re.search(r"\.name:\s+\S*dec_attn_pair_kernel\S*\n(.*?)\.wavefront_size", decode_asm, re.S)
    assert m
    meta = m.group(1)
    vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
    spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
    scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1))
    k = re.search(r"\.amdhsa_kernel \S*dec_attn_pair_kernel\S*\n(.*?)\.end_amdhsa_kernel", decode_asm, re.S)
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size\s+(\d+)", k.group(1)).group(1))
    assert vgpr <= 128 and spill == 0 and scratch == 0, (vgpr, spill, scratch)
    assert lds + 65536 <= 80 * 1024, lds          # static + the 64 KB of wo: two per 160 KB CU


@pytest.fixture(scope="module")
def step_asm(tmp_path_factory):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    from yourmt3_amd import build as B
    out = tmp_path_factory.mktemp("isa") / "dec_step.s"
    flags = [f for f in B.FLAGS if f not in ("-fPIC",)]
    cmd = [hipcc] + flags + ["-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "yourmt3_amd", "csrc", "dec_step.hip"),
                             "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


def test_step_kernel_keeps_the_counted_wait_and_fits_two_per_cu(step_asm):
    """dec_step_kernel (the option YMT3_STEP_KERNEL=1) carries the attention pair's hand-off inside a loop over the layers: the four agent-scope
    partial stores, then EXACTLY ten loads, then `s_waitcnt vmcnt(10)`, then the row's arrival -- with no drain of the store queue in between
    (hipcc inserted one when the self-attention half had two instantiations) and no flat-address-space access (its argument struct is re-read
    from the kernel-argument segment per layer; without the address-space round trip every access becomes a flat one).  512 workgroups must be
    resident: <= 128 VGPRs, no scratch, 76 KB of LDS."""
    body = _kernel_body(step_asm, "dec_step_kernel")
    stores = [i for i, l in enumerate(body) if l.startswith("buffer_store_dwordx4") and " sc1" in l]
    waits = [i for i, l in enumerate(body) if l.startswith("s_waitcnt vmcnt(10)")]
    groups = [w for w in waits if any(st < w and w - st < 80 for st in stores)]
    assert groups, "the hand-off waits with vmcnt(10) right behind the partial stores"
    w = groups[0]
    last_store = max(st for st in stores if st < w)
    four = [l for l in body[last_store - 3:last_store + 1] if l.startswith("buffer_store_dwordx4") and " sc1" in l]
    assert len(four) == 4, body[last_store - 5:last_store + 1]
    between = [l for l in body[last_store + 1:w] if VMEM.match(l) or l.startswith("s_waitcnt vmcnt")]
    assert len(between) == 10 and all(l.startswith("global_load") for l in between), between
    after = [l for l in body[w + 1:] if VMEM.match(l)]
    assert after[0].startswith("global_atomic_add"), after[:3]
    assert not any(l.startswith("flat_load") or l.startswith("flat_atomic") for l in body)
    m = re.search(r"\.name:\s+\S*dec_step_kernel\S*\n(.*?)\.wavefront_size", step_asm, re.S)
    meta = m.group(1)
    vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
    spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
    scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1))
    assert vgpr <= 128 and spill == 0 and scratch == 0, (vgpr, spill, scratch)


def _asm(tmp_path_factory, name):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    from yourmt3_amd import build as B
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    flags = [f for f in B.FLAGS if f not in ("-fPIC",)]
    cmd = [hipcc] + flags + ["-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "yourmt3_amd", "csrc", name + ".hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


def test_moe_chain_kernels_fit_one_workgroup_per_cu_without_scratch(tmp_path_factory):
    """moe_chain_kernel keeps every later stage's weights in registers from entry (249 VGPRs in the bf16 form): a spill would put scratch traffic on
    the chain's critical path, and more than 256 VGPRs (512 threads: two waves per SIMD) or 160 KB of LDS would make the launch fail.  Four
    instantiations: QKV / lm_head tail x bf16 / fp8."""
    asm = _asm(tmp_path_factory, "moe_chain")
    metas = re.findall(r"\.name:\s+(\S*moe_chain_kernel\S*)\n(.*?)\.wavefront_size", asm, re.S)
    assert len(metas) == 4, [m[0] for m in metas]
    for name, meta in metas:
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1))
        assert vgpr <= 256 and spill == 0 and scratch == 0, (name, vgpr, spill, scratch)
    # the dynamic LDS block is set by the launcher from these constants: stage 3's two 16-row strips per wave behind the reduction space
    src = open(os.path.join(ROOT, "yourmt3_amd", "csrc", "moe_chain.hip")).read()
    assert "static_assert(MOE_CHAIN_LDS_BF16 <= 160 * 1024 && MOE_CHAIN_LDS_FP8 <= 160 * 1024" in src
