"""VQ-VAE structure encoder on the HIP path (esm_amd/vqvae.py StructureTokenEncoder, csrc/structenc.hip, the bias
epilogues of csrc/gemm.hip and esmk_geom_attn_ex) through the C-ABI: kernel-level checks against torch fp32 / the CPU
oracle and `encode()` end to end against outputs of the imported reference (tests/golden/vqenc_*.npz).

Tolerances.  The kNN graph, the relative-position gather and the final projection + codebook search are fp32 /
integer work: exact (edges: wherever the reference's own order is determined, see tests/_golden.assert_knn_equal).
The two geometric blocks run their GEMMs in bf16 with fp32 accumulation, so the pre-quantisation vector z carries
bf16-level error (gate: max |dz| <= 0.02 * max |z| + 0.02) and a structure token may differ from the fp32 reference
only where the reference's own decision is within that error: every residue whose margin between the nearest and the
second nearest code (stored in the fixture) exceeds the bound 2 |dz|_2 |e_1 - e_2| must get the reference's token.
The reference's encoder does not run in bf16 at all (oracle/make_golden.py vqenc_case), so there is no reference
bf16 drift to put next to ours.
"""
import numpy as np
import pytest
import torch

from tests import _golden as G

pytestmark = pytest.mark.gpu
VQENC = ["vqenc_tiny", "vqenc_tiny_seqid_resindex", "vqenc_tiny_short", "vqenc_tiny_sparse", "vqenc_d1024_vh128"]


def _lib():
    from esm_amd import _lib as L
    from esm_amd.engine import _stream
    return L, L.lib(), _stream()


def _encoder(fx):
    from esm_amd.vqvae import StructureTokenEncoder
    m = StructureTokenEncoder(int(fx["d"]), 1, int(fx["VH"]), int(fx["NL"]), 128, int(fx["n_codes"]))
    m.load_state_dict(G.weights_for(fx), strict=True)
    return m.cuda()


@pytest.mark.parametrize("epi", ["bf16", "residual", "swiglu", "f32"])
@pytest.mark.parametrize("M,N,K", [(300, 384, 256), (2048, 1024, 384), (8192, 1024, 1024)])
def test_gemm_bias_epilogues(epi, M, N, K):
    """bias=True Linear layers: the bias is added in the epilogue of every variant (and of the split-K reduce)."""
    from esm_amd.engine import pack_weight
    L, lib, s = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g)
    acc = a.float() @ w.bfloat16().float().T + bias
    ad, bd = a.cuda(), bias.cuda()
    alpha = 0.75
    if epi == "swiglu":
        wp = pack_weight(w, "cuda", L.PACK_SWIGLU)
        want = torch.nn.functional.silu(acc[:, :N // 2]) * acc[:, N // 2:]
        out = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
        code = L.EPI_SWIGLU
    else:
        wp = pack_weight(w, "cuda")
        code = {"bf16": L.EPI_BF16, "residual": L.EPI_RESIDUAL, "f32": L.EPI_F32}[epi]
        if epi == "bf16":
            want, out = alpha * acc, torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        elif epi == "f32":
            want, out = acc, torch.empty(M, N, dtype=torch.float32, device="cuda")
        else:
            x0 = torch.randn(M, N, generator=g)
            want, out = x0 + alpha * acc, x0.cuda()
    ws = torch.empty(max(16, lib.esmk_gemm_workspace_bytes(M, N, K)), dtype=torch.uint8, device="cuda")
    L.check(lib.esmk_gemm_ws(L.ptr(ad), K, L.ptr(wp), M, N, K, code, L.ptr(bd), alpha, L.ptr(out), L.ptr(ws), ws.numel(), s))
    tol = 2e-2 if out.dtype == torch.bfloat16 else 2e-3
    np.testing.assert_allclose(out.float().cpu().numpy(), want.numpy(), atol=tol * float(want.abs().max()), rtol=0)


@pytest.mark.parametrize("name", VQENC)
def test_knn_edges_match_reference(name):
    fx = G.load(name)
    m = _encoder(fx)
    sid = G.t(fx["sequence_id"]).cuda() if "sequence_id" in fx else None
    edges, emask = m.find_knn_edges(G.t(fx["coords"]).cuda(), (~G.t(fx["attention_mask"])).cuda(),
                                    G.t(fx["affine_mask"]).cuda(), sequence_id=sid, knn=16)
    G.assert_knn_equal(edges.cpu(), emask.cpu(), fx)


@pytest.mark.parametrize("E,zf", [(16, 0), (16, 1), (11, 0), (40, 0)])
def test_geom_attn_neighbourhood_rows(E, zf):
    """esmk_geom_attn_ex on rows of <= 16 tokens (fused kernel) against the oracle, with mask_and_zero_frameless = 0
    (rows whose keys are all masked get uniform attention, frameless queries are not zeroed) and = 1 (ESM3's setting:
    frameless queries give zeros)."""
    from oracle import trunk_ref
    L, lib, s = _lib()
    g = torch.Generator().manual_seed(3)
    R, H = 37, 8  # E <= 16: fused small-row kernel; E = 40: the general kernel with the flag off
    M = R * E
    proj = torch.randn(R, E, 15 * H, generator=g)
    coords = torch.randn(R, E, 3, 3, generator=g) * 3
    coords[torch.rand(R, E, generator=g) < 0.3] = float("nan")
    coords[5] = float("nan")  # a neighbourhood without any frame
    rot, trans, amask = trunk_ref.build_frames(coords)
    seq = torch.randint(0, 2, (R, E), generator=g)
    chain = torch.zeros(R, E, dtype=torch.int64)
    w = {"distance_scale_per_head": torch.randn(H, generator=g) * 0.5, "rotation_scale_per_head": torch.randn(H, generator=g) * 0.5}

    # oracle: geometric_attention with identity projections, so that its projected vectors are LN(x) - which is
    # what the kernel is fed
    wd = dict(w)
    wd["s_norm.weight"] = torch.ones(15 * H)
    wd["proj.weight"] = torch.eye(15 * H)
    wd["out_proj.weight"] = torch.eye(3 * H)
    x = proj
    want = trunk_ref.geometric_attention(x, rot, trans, amask, seq, chain, wd, "", H, mask_and_zero_frameless=bool(zf))
    aff = torch.cat([rot.flatten(-2), trans], dim=-1).reshape(M, 12).contiguous().cuda()
    pd = trunk_ref.layer_norm(x, wd["s_norm.weight"]).reshape(M, 15 * H).contiguous().cuda()
    out = torch.empty(M, 64, dtype=torch.bfloat16, device="cuda")
    scratch = torch.empty(lib.esmk_geom_scratch_bytes(M, H), dtype=torch.uint8, device="cuda")
    row_cu = torch.arange(0, M + 1, E, dtype=torch.int32, device="cuda")
    dev = lambda t, dt: t.reshape(-1).to(dt).contiguous().cuda()
    am, sq, ch = dev(amask, torch.uint8), dev(seq, torch.int32), dev(chain, torch.int32)
    ds, rs = w["distance_scale_per_head"].cuda(), w["rotation_scale_per_head"].cuda()
    L.check(lib.esmk_geom_attn_ex(L.ptr(pd), L.ptr(aff), L.ptr(am), L.ptr(sq), L.ptr(ch), L.ptr(row_cu), R, E, L.ptr(ds),
                                  L.ptr(rs), L.ptr(out), 64, L.ptr(scratch), M, H, zf, s))
    got = out.float().cpu()[:, :3 * H].reshape(R, E, 3 * H)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-2 * float(want.abs().max()), rtol=0)
    if zf == 0:
        assert float(want[5].abs().max()) > 0  # the all-masked neighbourhood really exercises the uniform branch
    else:
        assert float(got[~amask].abs().max()) == 0.0
    assert float(out[:, 3 * H:].abs().max()) == 0.0


def test_vq_quantize_kernel():
    L, lib, s = _lib()
    g = torch.Generator().manual_seed(4)
    R, E, d, n_codes = 150, 16, 256, 1000
    x = torch.randn(R * E, d, generator=g)
    keep = torch.rand(R, generator=g) > 0.2
    w = torch.randn(128, d, generator=g) / 16
    b = torch.randn(128, generator=g) * 0.1
    codes = torch.randn(n_codes, 128, generator=g)
    x0 = x.view(R, E, d)[:, 0].masked_fill(~keep[:, None], 0)
    z = x0 @ w.T + b
    dist = (z ** 2).sum(1, keepdim=True) - 2 * z @ codes.T + (codes.T ** 2).sum(0, keepdim=True)
    want = dist.argmin(1)
    zd = torch.empty(R, 128, device="cuda")
    zq = torch.empty(R, 128, device="cuda")
    idx = torch.empty(R, dtype=torch.int64, device="cuda")
    t = [v.contiguous().cuda() for v in (x, keep.to(torch.uint8), w.t(), b, codes.t(), (codes ** 2).sum(-1))]
    L.check(lib.esmk_vq_quantize(L.ptr(t[0]), E * d, L.ptr(t[1]), L.ptr(t[2]), L.ptr(t[3]), L.ptr(t[4]), L.ptr(t[5]), R, d, 128,
                                 n_codes, L.ptr(zd), L.ptr(zq), L.ptr(idx), s))
    np.testing.assert_allclose(zd.cpu().numpy(), z.numpy(), atol=2e-5, rtol=1e-5)
    assert torch.equal(idx.cpu(), want)
    assert torch.equal(zq.cpu(), codes[want])


@pytest.mark.parametrize("name", VQENC)
def test_encode_matches_reference(name):
    fx = G.load(name)
    m = _encoder(fx)
    opt = lambda k: G.t(fx[k]).cuda() if k in fx else None  # noqa: E731
    zs = []
    zq, idx = m.encode(G.t(fx["coords"]).cuda(), attention_mask=opt("attention_mask"), sequence_id=opt("sequence_id"),
                       residue_index=opt("residue_index"), _z_out=zs)
    want_idx, want_z, amask = G.t(fx["indices"]), G.t(fx["z"]), G.t(fx["affine_mask"])
    assert zq.shape == want_z.shape and idx.shape == want_idx.shape and idx.dtype == torch.int64
    codes = G.weights_for(fx)["codebook.embeddings"]
    assert torch.equal(zq.cpu(), codes[idx.cpu()])  # z_q is the chosen code
    # residues without a frame: z is the bias alone -> the token is exact
    assert torch.equal(idx.cpu()[~amask], want_idx[~amask])
    z = zs[0].cpu()  # pre-quantisation vectors of our path (esmk_vq_quantize returns them on request)
    dz = (z - want_z)[amask]
    assert float(dz.abs().max()) <= 0.02 * float(want_z.abs().max()) + 0.02, float(dz.abs().max())
    # tokens: exact wherever the reference's margin exceeds what dz can move
    flat = want_z[amask]
    dist = (flat ** 2).sum(1, keepdim=True) - 2 * flat @ codes.T + (codes.T ** 2).sum(0, keepdim=True)
    top2 = dist.topk(2, dim=1, largest=False)
    gap = (codes[top2.indices[:, 0]] - codes[top2.indices[:, 1]]).norm(dim=-1)
    safe = (top2.values[:, 1] - top2.values[:, 0]) > 2 * dz.norm(dim=-1) * gap
    got, want = idx.cpu()[amask], want_idx[amask]
    assert torch.equal(got[safe], want[safe])
    assert float((got == want).float().mean()) >= 0.9, float((got == want).float().mean())
    assert float(safe.float().mean()) >= 0.5  # the gate is not vacuous


def test_encode_decode_roundtrip_shapes_and_determinism():
    """Long protein through both VQ-VAE halves: coordinates -> tokens -> coordinates (random weights: shapes, ranges,
    determinism, chunked execution == single pass)."""
    from esm_amd import synth
    from esm_amd.vqvae import StructureTokenDecoder, StructureTokenEncoder
    fx = G.load("vqenc_tiny")
    enc = _encoder(fx)
    coords = synth.synth_backbone(3, 300, seed=5, nan_frac=0.05, jitter=0.15).cuda()
    zq, idx = enc.encode(coords)
    zq2, idx2 = enc.encode(coords)
    assert torch.equal(idx, idx2) and torch.equal(zq, zq2)
    assert idx.shape == (3, 300) and int(idx.min()) >= 0 and int(idx.max()) < int(fx["n_codes"])
    old = StructureTokenEncoder.MAX_RESIDUES_PER_CALL
    try:
        StructureTokenEncoder.MAX_RESIDUES_PER_CALL = 128  # 900 residues in 8 passes
        _, idx3 = enc.encode(coords)
    finally:
        StructureTokenEncoder.MAX_RESIDUES_PER_CALL = old
    assert float((idx3 == idx).float().mean()) >= 0.98  # GEMM tile / split-K choice depends on the pass size
    dfx = G.load("vqdec_tiny")
    dec = StructureTokenDecoder(int(dfx["d"]), int(dfx["H"]), int(dfx["NL"]))
    dec.load_state_dict(G.weights_for(dfx), strict=True)
    dec = dec.cuda()
    tok = torch.full((3, 302), 4099, dtype=torch.int64, device="cuda")
    tok[:, 1:-1] = idx
    tok[:, 0], tok[:, -1] = 4098, 4097
    out = dec.decode(tok)
    assert out["bb_pred"].shape == (3, 302, 3, 3) and torch.isfinite(out["bb_pred"]).all()


def test_released_geometry_full_length_against_oracle():
    """d 1024, 128 vector heads, 4096 codes, one 512-residue protein against the CPU oracle (pinned to the reference by
    the fixtures above): z within the fixture gate, tokens equal wherever the margin exceeds what dz can move."""
    from esm_amd import synth
    from oracle import vqvae_ref
    fx = G.load("vqenc_d1024_vh128")
    w = G.weights_for(fx)
    m = _encoder(fx)
    coords = synth.synth_backbone(1, 512, seed=123, nan_frac=0.08, jitter=0.15)
    torch.set_num_threads(16)
    _, want_idx, want_z = vqvae_ref.encode(coords, w, int(fx["VH"]), int(fx["NL"]))
    zs = []
    _, idx = m.encode(coords.cuda(), _z_out=zs)
    amask = torch.isfinite(coords).all(-1).all(-1)
    dz = (zs[0].cpu() - want_z)[amask]
    assert float(dz.abs().max()) <= 0.02 * float(want_z.abs().max()) + 0.02, float(dz.abs().max())
    codes = w["codebook.embeddings"]
    flat = want_z[amask]
    dist = (flat ** 2).sum(1, keepdim=True) - 2 * flat @ codes.T + (codes.T ** 2).sum(0, keepdim=True)
    top2 = dist.topk(2, dim=1, largest=False)
    gap = (codes[top2.indices[:, 0]] - codes[top2.indices[:, 1]]).norm(dim=-1)
    safe = (top2.values[:, 1] - top2.values[:, 0]) > 2 * dz.norm(dim=-1) * gap
    got, want = idx.cpu()[amask], want_idx[amask]
    assert torch.equal(got[safe], want[safe])
    assert float((got == want).float().mean()) >= 0.9 and float(safe.float().mean()) >= 0.5
    assert torch.equal(idx.cpu()[~amask], want_idx[~amask])
