"""CPU-side tests: ABI surface, blob format, host tables, golden fixtures vs the oracle, sharding."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ymt3_oracle as O
from yourmt3_amd import tables
from yourmt3_amd.config import YMT3Config, CConfig, baseline_config
from yourmt3_amd.weights import make_weights, pack_blob, unpack_blob, f32_to_bf16_bits, bf16_bits_to_f32

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_builds_loads_and_exports_every_header_symbol():
    import __graft_entry__ as ge
    ge.build()
    from yourmt3_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "ymt3.h")).read()
    declared = set(re.findall(r"\b(ymt3_[a-z_0-9]+)\s*\(", header))
    declared -= {"ymt3_ctx"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.ymt3_abi_version() == 3


def test_cconfig_matches_header_field_order():
    header = open(os.path.join(ROOT, "include", "ymt3.h")).read()
    body = header[header.index("typedef struct ymt3_config {"):header.index("} ymt3_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for line in body.splitlines()[1:]:
        m = re.match(r"\s*(int32_t|float)\s+(.*);", line)
        if m:
            fields += [(f.strip(), m.group(1)) for f in m.group(2).split(",")]
    py = [(n, "float" if t is ctypes.c_float else "int32_t") for n, t in CConfig._fields_]
    assert fields == py
    assert ctypes.sizeof(CConfig) == 4 * len(py)


def test_no_gpu_means_a_loud_error_not_a_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from yourmt3_amd._lib import YMT3Error
    from yourmt3_amd.model import YourMT3
    with pytest.raises(YMT3Error):
        YourMT3(YMT3Config(segment_samples=8191))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "yourmt3_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_blob_roundtrip_and_bf16_rounding():
    cfg = YMT3Config(segment_samples=8191, max_decode_len=16)
    W = make_weights(cfg)
    W2 = unpack_blob(pack_blob({**W, **tables.derived_tables(W, cfg)}))
    for k, v in W.items():
        assert torch.equal(v, W2[k]), k
    assert W2["fe.mel_start"].dtype == torch.int32
    x = torch.tensor([1.0, 1.00390625, 1.005859375, -3.1415926, 65504.0, 1e-40])
    assert torch.equal(bf16_bits_to_f32(f32_to_bf16_bits(x)), x.bfloat16().float())   # RNE, ties to even


def test_bucket_tables_match_oracle_hf_and_golden():
    z = np.load(os.path.join(GOLD, "relpos_buckets.npz"))
    q = np.arange(512)[:, None]
    k = np.arange(512)[None, :]
    for key, bidir in (("bidirectional", True), ("causal", False)):
        got = tables.relative_position_bucket(k - q, bidir, 32, 128)
        assert np.array_equal(got, z[key].astype(np.int64))
        assert np.array_equal(got, O.relative_position_bucket(k - q, bidir, 32, 128))
    cfg = YMT3Config()
    W = make_weights(cfg)
    assert torch.equal(tables.encoder_bias_table(W["enc.relbias"], 256, cfg), O.encoder_bias_by_offset(W["enc.relbias"], 256, cfg))
    assert torch.equal(tables.decoder_bias_table(W["dec.relbias"], 1024, cfg), O.decoder_bias_by_distance(W["dec.relbias"], 1024, cfg))


def test_frontend_tables_reproduce_the_oracle_filterbank():
    cfg = YMT3Config()
    t = tables.frontend_tables(cfg)
    fb = O.mel_filterbank_htk(cfg.n_mels, cfg.n_fft, cfg.sample_rate, cfg.f_min, cfg.f_max)
    dense = torch.zeros_like(fb)
    for m in range(cfg.n_mels):
        s, n, o = int(t["fe.mel_start"][m]), int(t["fe.mel_len"][m]), int(t["fe.mel_off"][m])
        dense[m, s:s + n] = t["fe.mel_w"][o:o + n]
    assert torch.equal(dense, fb)
    assert torch.allclose(t["fe.window"], O.hann_window(cfg.n_fft), atol=1e-7)
    assert t["fe.tw"].shape == (cfg.n_fft // 2, 2) and t["fe.untw"].shape == (cfg.n_fft // 2 + 1, 2)


@pytest.mark.parametrize("name,cfg,n", [
    ("small_t64", YMT3Config(segment_samples=8191, max_decode_len=64), 48),
    ("mc3_t64", YMT3Config(segment_samples=8191, max_decode_len=32, n_channels=3), 24),
])
def test_oracle_reproduces_golden_fixture(name, cfg, n):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    W = make_weights(cfg, seed=int(z["seed_weights"]))
    a = O.synthetic_audio(2, cfg, seed=int(z["seed_audio"]))
    mel, enc = O.encode(a, W, cfg, True)
    assert np.allclose(mel.numpy(), z["mel"], atol=1e-5)
    assert np.array_equal(f32_to_bf16_bits(enc), z["enc_bf16"])
    toks = O.greedy_decode(enc, W, cfg, n, True)
    assert np.array_equal(toks.numpy(), z["tokens"])


def test_oracle_reproduces_the_full_length_golden_fixture():
    """tests/golden/full_t256_l1024.npz: configs[1] shapes, 2 segments x 1024 steps, ids + margins + logits at late positions
    (127 | 382, 383, 384 | 767 | 1023); its encoder output is the one stored in full_t256.npz."""
    z = np.load(os.path.join(GOLD, "full_t256_l1024.npz"))
    base = np.load(os.path.join(GOLD, str(z["enc_from"]) + ".npz"))
    cfg = YMT3Config(max_decode_len=1024, eos_id=-1)
    W = make_weights(cfg, seed=int(z["seed_weights"]))
    enc = bf16_bits_to_f32(base["enc_bf16"]).view(2, cfg.n_frames, cfg.d_model)
    toks, logits = O.greedy_decode(enc, W, cfg, int(z["n_steps"]), True, return_logits=True)
    assert np.array_equal(toks.numpy(), z["tokens"])
    steps = z["logit_steps"].tolist()
    assert steps == [0, 127, 382, 383, 384, 767, 1023]
    assert np.allclose(logits[:, :, steps, :].numpy(), z["logits"], atol=1e-5)
    top2 = logits.topk(2, -1).values
    assert np.allclose((top2[..., 0] - top2[..., 1]).numpy(), z["margin"], atol=1e-5)


def test_oracle_moe_routing_can_be_teacher_forced():
    """The hooks the MoE GPU parity tests use: forcing the oracle's own expert choices back into it changes nothing (deficit 0);
    a choice outside its top two shows up as a positive deficit and moves the logits."""
    from yourmt3_amd.config import FFN_MOE
    cfg = YMT3Config(segment_samples=8191, max_decode_len=16, dec_ffn=FFN_MOE, eos_id=-1)
    W = make_weights(cfg)
    _, enc = O.encode(O.synthetic_audio(2, cfg), W, cfg, True)
    O.MOE_CHOSEN = []
    try:
        t0, l0 = O.greedy_decode(enc, W, cfg, 6, True, return_logits=True)
        own = O.MOE_CHOSEN
    finally:
        O.MOE_CHOSEN = None
    assert len(own) == 6 * cfg.n_dec_layers and own[0].shape == (2, 2)

    def forced_run(choices):
        O.MOE_FORCED_SEL, O.MOE_FORCED_DEFICIT = iter(choices), []
        try:
            t, l = O.greedy_decode(enc, W, cfg, 6, True, return_logits=True)
            return t, l, torch.stack(O.MOE_FORCED_DEFICIT)
        finally:
            O.MOE_FORCED_SEL = O.MOE_FORCED_DEFICIT = None

    t1, l1, d1 = forced_run(own)
    assert torch.equal(t1, t0) and torch.equal(l1, l0) and float(d1.max()) == 0.0
    other = [c.clone() for c in own]
    other[3][0, 1] = (set(range(cfg.n_experts)) - set(other[3][0].tolist())).pop()
    _, l2, d2 = forced_run(other)
    assert float(d2.max()) > 0.0 and int((d2 > 0).sum()) == 1 and float((l2 - l0).abs().max()) > 1e-3


def test_perceiver_tf_oracle_shape_mixing_and_blob():
    """a9 oracle (build-defined spec): (B, T, F') -> latents (B, T, K, D) -> (B, T, d_model); K is free of T; the spectral
    cross-attention sees one frame, the temporal transformer mixes frames; its attention primitive equals torch's SDPA with
    scale 1; the blob carries every ptf tensor plus the derived temporal bias table."""
    import torch
    from oracle import ymt3_oracle as O
    from oracle.perceiver_oracle import encoder_perceiver_tf, perceiver_tf_latents, _mha
    from yourmt3_amd.config import ENC_PERCEIVER_TF
    from yourmt3_amd.tables import derived_tables
    from yourmt3_amd.weights import make_weights, pack_blob, unpack_blob
    cfg = YMT3Config(segment_samples=8191, encoder_type=ENC_PERCEIVER_TF, n_latents=32, n_enc_layers=0, ptf_blocks=2)
    W = make_weights(cfg)
    mel = O.logmel(O.synthetic_audio(2, cfg), cfg)
    z = perceiver_tf_latents(mel, W, cfg, False)
    assert z.shape == (2, cfg.n_frames, 32, cfg.ptf_d)
    enc = encoder_perceiver_tf(mel, W, cfg, False)
    assert enc.shape == (2, cfg.n_frames, cfg.d_model) and torch.isfinite(enc).all()
    assert torch.allclose(encoder_perceiver_tf(mel[1:], W, cfg, False)[0], enc[1], atol=1e-5)       # segments are independent
    mel2 = mel.clone(); mel2[0, 0] += 1.0                                                             # perturb frame 0 of segment 0
    z2 = perceiver_tf_latents(mel2, W, cfg, False)
    assert (z2[0, -1] - z[0, -1]).abs().max() > 1e-4 and torch.equal(z2[1], z[1])                    # reaches the last frame; not segment 1
    one = cfg.with_(ptf_blocks=1)
    W1 = make_weights(one)
    assert (encoder_perceiver_tf(mel, W1, one, True) - encoder_perceiver_tf(mel, W1, one, False)).abs().mean() < 0.02
    big = cfg.with_(n_latents=64)
    assert perceiver_tf_latents(mel, make_weights(big), big, False).shape == (2, cfg.n_frames, 64, cfg.ptf_d)
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(3, 16, 128, generator=g) for _ in range(3))
    sd = torch.nn.functional.scaled_dot_product_attention
    ref = sd(q.view(3, 16, 2, 64).transpose(1, 2), k.view(3, 16, 2, 64).transpose(1, 2), v.view(3, 16, 2, 64).transpose(1, 2), scale=1.0)
    assert torch.allclose(_mha(q, k, v, None, False), ref.transpose(1, 2).reshape(3, 16, 128), atol=1e-5)
    full = {**W, **derived_tables(W, cfg)}
    back = unpack_blob(pack_blob(full))
    assert back["ptf.bias_off"].shape == (cfg.ptf_d // 64, 2 * cfg.n_frames - 1) and back["ptf.1.t.wqkv"].shape == (3 * cfg.ptf_d, cfg.ptf_d)
    assert back["ptf.out_w"].shape == (cfg.d_model, 32 * cfg.ptf_d)


def test_baseline_configs():
    assert baseline_config(1).eos_id == -1 and baseline_config(1).max_decode_len == 1024
    assert baseline_config(3).n_channels == 13 and baseline_config(3).max_decode_len == 256
    assert baseline_config(0).n_frames == 256 and abs(baseline_config(0).segment_seconds - 2.048) < 1e-9


def test_shard_ranges_cover_every_segment_once():
    from yourmt3_amd.dist import shard_range, shard_sizes
    for n in (0, 1, 7, 64, 65, 512):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert sum(shard_sizes(n, w)) == n


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from yourmt3_amd.dist import init_distributed, shard_range, all_gather_tokens, gather_floats
rank, world, _ = init_distributed(int(os.environ["WORLD_SIZE"]))
n = int(sys.argv[2])
lo, hi = shard_range(n, rank, world)
full = (torch.arange(n * 3 * 5, dtype=torch.int32).view(n, 3, 5) * 7) % 1000
out = all_gather_tokens(full[lo:hi].clone(), world, n_segments=n)
assert torch.equal(out, full), (rank, out.shape)
stats = gather_floats([rank + 0.5, 10.0 * rank], world)          # bench.py's per-rank times travel this way
assert stats == [[r + 0.5, 10.0 * r] for r in range(world)], stats
torch.distributed.barrier()
print("rank", rank, "ok")
"""


def _free_port():
    """A port nobody is listening on right now (fixed ports collided with earlier runs' sockets in TIME_WAIT once in a while)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


@pytest.mark.parametrize("n", [8, 7])
def test_all_gather_world_size_2_gloo(tmp_path, n):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(n)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]       # (a cold `import torch` in every rank can take minutes on a freshly built tree)
    assert all(p.returncode == 0 for p in procs), outs


@pytest.mark.parametrize("n", [61, 5])
def test_all_gather_world_size_8_gloo_ragged(tmp_path, n):
    """The driver's 8-GPU shape on CPU: eight ranks, ragged shards (61 segments: seven ranks of 8 and one of 5; 5 segments: three
    ranks own nothing and contribute padding only), ids back in segment order on every rank, per-rank stats gathered."""
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="8", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(n)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(8)]
    outs = [p.communicate(timeout=1500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_launcher_starts_one_rank_per_gpu_and_gathers(tmp_path):
    """`python bench.py --gpus N` without WORLD_SIZE becomes the launcher (yourmt3_amd.dist.launch_local_ranks): N fresh
    children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set.  Here the children are the gloo worker above."""
    from yourmt3_amd.dist import launch_local_ranks
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    code = ("import sys; sys.path.insert(0, %r); from yourmt3_amd.dist import launch_local_ranks; "
            "sys.exit(launch_local_ranks(2, [sys.executable, %r, %r, '7'], timeout=850))" % (ROOT, str(script), ROOT))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(l for l in r.stdout.splitlines() if l.startswith("rank")) == ["rank 0 ok", "rank 1 ok"]
    # a failing rank ends the job with its exit code instead of leaving the others in a collective
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(60)\n")
    t0 = __import__("time").time()
    assert launch_local_ranks(2, [sys.executable, str(bad)], timeout=50) == 3
    assert __import__("time").time() - t0 < 30


def test_bench_refuses_a_rank_count_mismatch_and_never_runs_without_a_gpu():
    """No silent 1-rank run when N ranks were asked for; and with no GPU the ranks fail loudly (no CPU fallback)."""
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    import torch
    if not torch.cuda.is_available():
        env.pop("WORLD_SIZE", None)
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=dict(env, YMT3_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
        assert r.returncode != 0 and "GPU" in (r.stdout + r.stderr) and '"metric"' not in r.stdout


def test_header_is_plain_c(tmp_path):
    """include/ymt3.h is the boundary a C, Go (cgo) or Java (JNI) host would bind: it must compile as strict C99."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "ymt3.h"\nint main(void) { ymt3_config c; int (*f)(void) = ymt3_abi_version; (void)c; (void)f; return 0; }\n')
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
