"""Exploratory parity probe on the GPU box (not a test): prints max diffs per stage vs the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yourmt3_amd.config import YMT3Config
from yourmt3_amd.model import YourMT3
from yourmt3_amd.weights import make_weights
from oracle import ymt3_oracle as O

def probe(cfg, B, n_steps):
    print("=== cfg frames", cfg.n_frames, "B", B, "steps", n_steps, flush=True)
    W = make_weights(cfg)
    m = YourMT3(cfg, W, max_batch=max(B, 2))
    print("device MB", m.device_bytes / 1e6)
    a = O.synthetic_audio(B, cfg)
    mel_o = O.logmel(a, cfg)
    mel_g = m.logmel(a.cuda()).cpu()
    print("logmel max abs diff", (mel_o - mel_g).abs().max().item(), "mean", (mel_o - mel_g).abs().mean().item(), flush=True)
    # gemm
    g = torch.Generator().manual_seed(3)
    A = torch.randn(200, 512, generator=g).bfloat16(); Wt = torch.randn(256, 512, generator=g).bfloat16()
    c = m.test_gemm(A.cuda(), Wt.cuda()).cpu()
    ref = A.float() @ Wt.float().T
    print("gemm max rel", ((c - ref).abs().max() / ref.abs().max()).item(), flush=True)
    h0 = O.input_projection(mel_o, W, True)
    enc_o = O.encoder_t5(h0, W, cfg, True)
    enc_g = m.encode(mel_o.cuda()).float().cpu()
    d = (enc_o - enc_g).abs()
    print("encoder max abs diff", d.max().item(), "mean", d.mean().item(), "ref rms", enc_o.pow(2).mean().sqrt().item(), flush=True)
    toks_o, lg_o = O.greedy_decode(enc_o, W, cfg, n_steps, True, return_logits=True)
    toks_g, lg_g = m.decode(enc_o.bfloat16().cuda(), n_steps, forced=toks_o.cuda(), return_logits=True)
    lg_g = lg_g.cpu(); toks_g = toks_g.cpu()
    dl = (lg_o - lg_g).abs()
    print("teacher-forced logits max abs diff", dl.max().item(), "mean", dl.mean().item(), "std", lg_o.std().item())
    print("per-step max diff", [round(x, 4) for x in dl.amax(dim=(0, 1, 3)).tolist()][:16])
    top2 = lg_o.topk(2, -1).values; margin = top2[..., 0] - top2[..., 1]
    mism = (toks_o != toks_g)
    print("forced argmax mismatches", mism.sum().item(), "of", mism.numel(), "margins at mismatch", margin[mism].tolist()[:8])
    toks_f = m.decode(enc_o.bfloat16().cuda(), n_steps).cpu()
    eq = (toks_f == toks_o)
    first = [int((~eq[b, k]).nonzero()[0]) if (~eq[b, k]).any() else n_steps for b in range(B) for k in range(cfg.n_channels)]
    print("free-running: first divergence per row", first, "min margin", margin.min().item())
    toks_e = m.inference(a.cuda(), max_token_length=n_steps).cpu()
    print("e2e vs decode-from-oracle-enc equal:", torch.equal(toks_e, toks_f), flush=True)
    print(toks_f[0, 0, :24].tolist())
    print(toks_o[0, 0, :24].tolist())
    m.close()

probe(YMT3Config(segment_samples=8191, max_decode_len=64), 2, 48)
probe(YMT3Config(max_decode_len=128), 2, 96)
