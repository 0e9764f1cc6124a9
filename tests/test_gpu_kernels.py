"""GPU parity of the HIP building blocks, called through the C-ABI
(``tvc_gemm_bf16``, ``tvc_layernorm``, ``tvc_attention``, ``tvc_cosine_matrix``).
Floating-point kernels: compared with a plain PyTorch fp32 restatement of the
same op on the same (bf16-rounded) operands; tolerances stated per test."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@pytest.mark.parametrize("I,J,K", [(256, 256, 64), (512, 300, 128), (768, 1000, 640), (100, 77, 64),
                                   (1024, 514, 1024), (4, 1, 64)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_epilogues(gpu_engine, I, J, K, epi):
    a = _rand((I, K), 1, K ** -0.5).to(torch.bfloat16)
    b = _rand((J, K), 2).to(torch.bfloat16)
    bias = _rand((I,), 3, 0.1)
    ref = b.float() @ a.float().t() + bias          # [J, I]
    dev = "cuda:0"
    if epi == 3:
        base = _rand((J, I), 4)
        out = base.clone().to(dev)
        gpu_engine.gemm(a.to(dev), b.to(dev), bias.to(dev), 3, out=out)
        ref = base + ref
    else:
        out = gpu_engine.gemm(a.to(dev), b.to(dev), bias.to(dev), epi)
    if epi == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    got = out.float().cpu()
    # fp32 accumulate of exact bf16 products: 1e-4 relative to the row scale;
    # bf16 outputs add one rounding (2^-9 relative)
    tol = 2e-4 * (1 + ref.abs().max().item()) if epi in (0, 3) else 1e-2 * (1 + ref.abs().max().item())
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < tol


@pytest.mark.parametrize("I,J,K,epi", [(3072, 12800, 768, 2), (1024, 8200, 256, 1), (2304, 20000, 512, 1)])
def test_gemm_persistent_many_tiles(gpu_engine, I, J, K, epi):
    """>= 512 tiles: the persistent ring kernel with several tiles (and several bias slices)
    per workgroup; every row is checked (a stale bias slot shows up only under this load)."""
    g = torch.Generator(device="cuda:0").manual_seed(7)
    a = (torch.randn(I, K, device="cuda:0", generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(J, K, device="cuda:0", generator=g).to(torch.bfloat16)
    bias = torch.randn(I, device="cuda:0", generator=g) * 3.0          # large: a wrong bias slice is O(1) wrong
    out = gpu_engine.gemm(a, b, bias, epi).float()
    ref = b.float() @ a.float().t() + bias
    if epi == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    assert (out - ref).abs().max().item() < 1e-2 * (1 + ref.abs().max().item())


@pytest.mark.parametrize("I,J,K,epi", [(1024, 131584, 256, 1), (1024, 66000, 512, 2), (768, 87300, 128, 0)])
def test_gemm_tile_counts_just_above_a_round(gpu_engine, I, J, K, epi):
    """Tile counts just above a multiple of 256 (514 x 4 = 2056, ...): a last, nearly empty round of
    the persistent kernel (or, with TVC_GEMM_SPLITK_TAIL=1, the split-K partial + finish kernels for the
    left-over tile columns); every output row is compared, the tail rows separately."""
    g = torch.Generator(device="cuda:0").manual_seed(9)
    a = (torch.randn(I, K, device="cuda:0", generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(J, K, device="cuda:0", generator=g).to(torch.bfloat16)
    bias = torch.randn(I, device="cuda:0", generator=g) * 3.0
    out = gpu_engine.gemm(a, b, bias, epi).float()
    nIt = (I + 255) // 256
    jt_full = ((nIt * ((J + 255) // 256)) // 256 * 256) // nIt
    for lo, hi in ((0, jt_full * 256), (jt_full * 256, J)):
        ref = b[lo:hi].float() @ a.float().t() + bias
        if epi == 2:
            ref = ref * torch.sigmoid(1.702 * ref)
        tol = (2e-4 if epi == 0 else 1e-2) * (1 + ref.abs().max().item())
        assert (out[lo:hi] - ref).abs().max().item() < tol


@pytest.mark.parametrize("I,J,K,epi", [(1024, 20000, 64, 1), (512, 40000, 128, 0), (768, 30000, 192, 2)])
def test_gemm_persistent_without_bias_and_short_k(gpu_engine, I, J, K, epi):
    """The persistent kernel with bias = nullptr (its epilogue keeps the bias vectors in registers: zeros here) and with
    one to three K-tiles per output tile (every K-tile is then a tile's FIRST one, whose counted waits let the previous
    epilogue's stores pass: 16 of them for bf16 outputs, 32 for fp32)."""
    g = torch.Generator(device="cuda:0").manual_seed(13)
    a = (torch.randn(I, K, device="cuda:0", generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(J, K, device="cuda:0", generator=g).to(torch.bfloat16)
    out = gpu_engine.gemm(a, b, None, epi).float()
    assert torch.equal(out, gpu_engine.gemm(a, b, None, epi).float())
    ref = b.float() @ a.float().t()
    if epi == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    tol = (2e-4 if epi == 0 else 1e-2) * (1 + ref.abs().max().item())
    assert (out - ref).abs().max().item() < tol


def test_gemm_identity_asymmetric(gpu_engine):
    """A = I with an asymmetric B catches a transposed accumulator map."""
    K = 256
    a = torch.eye(K).to(torch.bfloat16)
    b = (torch.arange(300 * K).reshape(300, K) % 251).float().to(torch.bfloat16)
    out = gpu_engine.gemm(a.cuda(), b.cuda(), None, 0).cpu()
    assert torch.equal(out, b.float())


@pytest.mark.parametrize("rows,d", [(5, 128), (1000, 256), (257, 768), (514, 1024)])
def test_layernorm(gpu_engine, rows, d):
    x = _rand((rows, d), 5, 3.0) + 0.5
    g = 1 + _rand((d,), 6, 0.1)
    b = _rand((d,), 7, 0.1)
    ref = torch.nn.functional.layer_norm(x, (d,), g, b, 1e-5)
    got = gpu_engine.layernorm(x.cuda(), g.cuda(), b.cuda()).float().cpu()
    # output is bf16: half an ulp of |ref| <= 2^-9 * max
    assert (got - ref).abs().max().item() < 2 ** -8 * (1 + ref.abs().max().item())


def _attn_ref(qkv, n_seq, T, heads, causal):
    d = heads * 64
    q, k, v = qkv.float().view(n_seq, T, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.full((T, T), float("-inf")).triu(1)
    o = s.softmax(-1) @ v
    return o.permute(0, 2, 1, 3).reshape(n_seq * T, d)


@pytest.mark.parametrize("n_seq,T,heads,causal", [(3, 17, 4, False), (2, 50, 12, False), (5, 77, 2, True),
                                                  (4, 77, 8, True), (2, 257, 16, False), (3, 20, 2, True),
                                                  (1, 1, 1, False), (2, 288, 2, False), (2, 33, 2, True)])
def test_attention(gpu_engine, n_seq, T, heads, causal):
    qkv = _rand((n_seq * T, 3 * heads * 64), 8, 1.0).to(torch.bfloat16)
    ref = _attn_ref(qkv, n_seq, T, heads, causal)
    got = gpu_engine.attention(qkv.cuda(), n_seq, T, heads, causal).float().cpu()
    assert torch.isfinite(got).all()
    # P and O are rounded to bf16: |err| <~ 2^-8 * max|v|
    assert (got - ref).abs().max().item() < 3e-2
    assert (got - ref).abs().mean().item() < 3e-3


@pytest.mark.parametrize("causal", [False, True])
def test_attention_length_sweep(gpu_engine, causal):
    """Every tile / pair boundary of the sequence length (16 n, 16 n + 1, 32 n, 32 n + 1, the template switches at 32 / 64 /
    96 tokens, the 257-token form's neighbours, the 288-token maximum): the fills clamp their row index and select zeros at
    the LDS write, an unpaired last tile multiplies its own V rows by zeros, stores are 16-byte pieces after a lane swap."""
    n_seq, heads = 3, 2
    for T in (2, 15, 16, 17, 31, 32, 33, 48, 49, 63, 64, 65, 80, 96, 97, 128, 255, 256, 257, 258, 271, 272, 273, 288):
        qkv = _rand((n_seq * T, 3 * heads * 64), 100 + T, 1.0).to(torch.bfloat16)
        ref = _attn_ref(qkv, n_seq, T, heads, causal)
        got = gpu_engine.attention(qkv.cuda(), n_seq, T, heads, causal).float().cpu()
        assert torch.isfinite(got).all(), T
        assert (got - ref).abs().max().item() < 3e-2, (T, (got - ref).abs().max().item())
        assert (got - ref).abs().mean().item() < 3e-3, T


def test_attention_nan_stays_inside_its_sequence_and_head(gpu_engine):
    """A NaN in one (sequence, head)'s keys / values / queries makes THAT item's outputs NaN and leaves every other item
    bit-identical (the kernels clamp row indices and multiply an unpaired last tile's own V rows by zero probabilities:
    nothing of one item is ever read for another)."""
    n_seq, T, heads = 4, 257, 3
    qkv = _rand((n_seq * T, 3 * heads * 64), 21, 1.0).to(torch.bfloat16)
    clean = gpu_engine.attention(qkv.cuda(), n_seq, T, heads, False).cpu()
    bad = qkv.clone()
    W = heads * 64
    bad[1 * T + 256, 2 * W + 1 * 64 + 5] = float("nan")        # sequence 1, head 1: the LAST value row (the unpaired tile)
    bad[2 * T + 7, 1 * W + 2 * 64 + 9] = float("nan")          # sequence 2, head 2: a key
    got = gpu_engine.attention(bad.cuda(), n_seq, T, heads, False).cpu()
    o = got.view(n_seq, T, heads, 64); c = clean.view(n_seq, T, heads, 64)
    hit = torch.zeros((n_seq, heads), dtype=torch.bool); hit[1, 1] = True; hit[2, 2] = True
    for s_ in range(n_seq):
        for h_ in range(heads):
            if hit[s_, h_]:
                assert torch.isnan(o[s_, :, h_].float()).any()
            else:
                assert torch.equal(o[s_, :, h_].view(torch.int16), c[s_, :, h_].view(torch.int16)), (s_, h_)


def test_attention_spiky_scores(gpu_engine):
    """Large score range (one dominant key per query) must not overflow."""
    n_seq, T, heads = 2, 257, 2
    qkv = _rand((n_seq * T, 3 * heads * 64), 9, 1.0)
    qkv[:, :heads * 64] *= 8.0
    qkv = qkv.to(torch.bfloat16)
    ref = _attn_ref(qkv, n_seq, T, heads, False)
    got = gpu_engine.attention(qkv.cuda(), n_seq, T, heads, False).float().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 6e-2


@pytest.mark.parametrize("N,M,D", [(7, 33, 128), (300, 1000, 512), (48, 20, 768)])
def test_cosine_matrix(gpu_engine, pkg, N, M, D):
    import ctypes as C
    x = _rand((N, D), 10, 2.0)
    y = _rand((M, D), 11, 0.5)
    xn = x.double() / x.double().norm(dim=1, keepdim=True)
    yn = y.double() / y.double().norm(dim=1, keepdim=True)
    ref = xn @ yn.t()      # src/utils/metrics.py:162-164
    out = torch.empty((N, M), dtype=torch.float32, device="cuda:0")
    xd, yd = x.cuda(), y.cuda()
    rc = gpu_engine.lib.tvc_cosine_matrix(gpu_engine.handle, C.c_void_p(xd.data_ptr()), N, C.c_void_p(yd.data_ptr()),
                                          M, D, C.c_void_p(out.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    # split-bf16 (3 products) + fp32 accumulate: 1e-5 absolute on cosines (bar: 1e-4)
    assert (out.double().cpu() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("env", [{"TVC_GEMM_SPLITK_TAIL": "1"}, {"TVC_GEMM_SPLITK_SMALL": "1"}, {"TVC_GEMM_RING_FORM": "1"},
                                 {"TVC_GEMM_VARIANT": "0"}])
def test_gemm_variants_in_subprocess(env):
    """Env switches read once per process: TVC_GEMM_SPLITK_TAIL=1 the split-K tail for left-over tile columns,
    TVC_GEMM_SPLITK_SMALL=1 split-K for GEMMs of a few tiles (latency mode), TVC_GEMM_RING_FORM=1 the general ring
    form everywhere, TVC_GEMM_VARIANT=0 the one-tile-per-workgroup kernel everywhere.  Same outputs as the PyTorch
    restatement.  (Ring forms 2 / 3 and the four-wave kernel were removed in round 3.)"""
    import os
    import subprocess
    import sys
    code = r'''
import torch, tvc_amd as pkg
eng = pkg.TVCEngine()
g = torch.Generator(device="cuda:0").manual_seed(11)
for I, J, K, epi in ((1024, 2048, 1024, 1), (768, 3000, 640, 2), (3072, 12800, 768, 2), (256, 2304, 4096, 1),
                     (1024, 131584, 256, 1), (1024, 66000, 512, 2), (1024, 257, 4096, 1), (3072, 300, 1024, 2),
                     (768, 77, 768, 1)):
    a = (torch.randn(I, K, device="cuda:0", generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(J, K, device="cuda:0", generator=g).to(torch.bfloat16)
    bias = torch.randn(I, device="cuda:0", generator=g) * 3.0
    out = eng.gemm(a, b, bias, epi).float()
    ref = b.float() @ a.float().t() + bias
    if epi == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    err = (out - ref).abs().max().item()
    assert err < 1e-2 * (1 + ref.abs().max().item()), (I, J, K, epi, err)
print("SOLO_OK")
'''
    env = dict(os.environ, **env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SOLO_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_ring_forms_are_bit_identical():
    """every ring form sums each output element over K in the same order and runs the same epilogue
    arithmetic, so forms 1 and 4 (the default) return the same BITS on tower-sized launches (incl. QuickGELU, a
    ragged number of tile rounds, K = 64 and the fp32 epilogue).  scripts/gemm_form_check.py prints a checksum of the
    raw output bits per shape; the env switch is read once per process, hence the subprocesses."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = {}
    for form in ("1", "4"):
        env = dict(os.environ, TVC_GEMM_RING_FORM=form)
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gemm_form_check.py")], cwd=root, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "FORM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        sums[form] = [ln.split("checksum")[1].strip() for ln in r.stdout.splitlines() if "checksum" in ln]
        assert len(sums[form]) == 8
    assert sums["1"] == sums["4"], sums


def test_bank_filter_ring_and_one_tile_loops_agree():
    """The filter pass of the bank search on GEMM form 4 (default) and on the one-tile-at-a-time loop (TVC_BANK_RING=0)
    sum the same products in the same order: identical top-k indices and similarities, for a ragged bank (R % 256 != 0),
    a ragged query count (padded query rows of the workspace are read but never listed), both bank dtypes, and one / two
    K-tiles per bank tile (D = 64, 128)."""
    import os
    import subprocess
    import sys
    code = r'''
import torch, tvc_amd as pkg
eng = pkg.TVCEngine()
g = torch.Generator(device="cuda:0").manual_seed(5)
for R, M, D, k, dt in ((70001, 300, 768, 5, torch.bfloat16), (33000, 512, 512, 10, torch.float32), (4096, 17, 1024, 3, torch.bfloat16),
                       (5000, 40, 64, 4, torch.bfloat16), (2500, 260, 128, 2, torch.float32)):
    bank = torch.nn.functional.normalize(torch.randn(R, D, device="cuda:0", generator=g), dim=-1).to(dt)
    q = torch.nn.functional.normalize(torch.randn(M, D, device="cuda:0", generator=g), dim=-1)
    eng.set_bank(bank)
    idx, sim = eng.bank_search(q, k, want_moments=False)[:2]        # no moments: the filter form
    ref = (q.double() @ bank.double().t()).topk(k, dim=1)
    if dt == torch.bfloat16:          # exact products; an fp32 bank's near-ties may order differently in fp64
        assert torch.equal(idx.long().cpu(), ref.indices.cpu()), (R, M, D)
    assert (sim.double().cpu() - ref.values.cpu()).abs().max().item() < 2e-6, (R, M, D)
    print("TOPK", R, M, int(idx.long().sum().item()), sim.cpu().numpy().tobytes().hex()[:64], float(sim.double().sum().item()).hex())
print("BANK_OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for ring in ("1", "0"):
        env = dict(os.environ, TVC_BANK_RING=ring)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "BANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        outs[ring] = [ln for ln in r.stdout.splitlines() if ln.startswith("TOPK")]
        assert len(outs[ring]) == 5
    assert outs["1"] == outs["0"], outs
