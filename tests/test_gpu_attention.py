"""Row f4: the global attention of PointTransformerV3 (models/PointTransformerV3.py:64-117) through the C ABI."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def sdpa_reference(qkv, H):
    """The reference's lines :96-113 in fp32 on the same (bf16-rounded) projections."""
    B, N, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.float().reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4).unbind(0)
    x = F.scaled_dot_product_attention(q, k, v)
    return x.transpose(1, 2).reshape(B, N, C)


@pytest.mark.parametrize("B,N,H,D", [(1, 256, 2, 192), (2, 1000, 2, 192), (1, 77, 1, 64), (2, 513, 3, 128), (1, 4096, 2, 192),
                                     (1, 300, 1, 256)])
def test_attention_matches_sdpa(B, N, H, D):
    from pointcloud_bridge_amd import ops
    torch.manual_seed(N + D)
    qkv = (torch.randn(B, N, 3 * H * D, device="cuda") * 0.7).to(torch.bfloat16)
    got = ops.attention(qkv, H).float()
    ref = sdpa_reference(qkv, H)
    assert got.shape == ref.shape
    # bf16 products with fp32 accumulation and an fp32 softmax; P is rounded to bf16 for the second product
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < 2e-2, float(err)
    assert float((got - ref).abs().mean() / ref.abs().mean()) < 6e-3


def test_attention_peaked_rows_and_scale():
    """rows whose softmax is (almost) one-hot, a custom scale, and large logits (no overflow: running max)"""
    from pointcloud_bridge_amd import ops
    torch.manual_seed(0)
    B, N, H, D = 1, 640, 2, 192
    qkv = torch.randn(B, N, 3, H, D, device="cuda")
    qkv[:, :, 0] *= 6.0
    qkv[:, :, 1] *= 6.0
    qkv = qkv.reshape(B, N, 3 * H * D).to(torch.bfloat16)
    got = ops.attention(qkv, H, scale=0.25).float()
    q, k, v = qkv.float().reshape(B, N, 3, H, D).permute(2, 0, 3, 1, 4).unbind(0)
    ref = F.scaled_dot_product_attention(q, k, v, scale=0.25).transpose(1, 2).reshape(B, N, H * D)
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-2


def test_attention_argument_checks():
    from pointcloud_bridge_amd import ops
    with pytest.raises(ValueError):
        ops.attention(torch.zeros(1, 8, 3 * 2 * 80, device="cuda"), 2)      # head_dim 80
    with pytest.raises(RuntimeError):
        ops.attention(torch.zeros(1, 8, 3 * 64), 1)                            # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        ops.attention(torch.zeros(1, 8, 3 * 64, device="cuda", requires_grad=True), 1)


def _ptv3(fixture):
    from pointcloud_bridge_amd.models.PointTransformerV3 import PointTransformerV3
    torch.manual_seed(int(fixture["init_seed"]))
    model = PointTransformerV3(num_classes=5, d_in=6, embed_dim=384, depth=int(fixture["depth"]), num_heads=2, mlp_ratio=4.,
                               qkv_bias=True, drop_rate=0.1, attn_drop_rate=0.1).cuda().eval()
    assert len(model.state_dict()) == int(fixture["num_state_keys"])
    with torch.no_grad():
        model.head[1].running_mean.copy_(torch.from_numpy(fixture["head_running_mean"]))
        model.head[1].running_var.copy_(torch.from_numpy(fixture["head_running_var"]))
    return model


def test_ptv3_logits_against_the_reference_fixture(monkeypatch):
    """PointTransformerV3 in the configuration of inference_ptv3.py:101-105 (depth 3 in the fixture) against eval
    logits of the REFERENCE on the CPU (tests/golden/make_golden_ptv3.py).  fp32 mode (the reference's composition
    on ATen, on the GPU): 1e-4.  bf16 mode (bf16 token pipeline, attention on csrc/attention.hip): bar measured on
    this fixture and written here; the test also checks that the library kernel is what ran."""
    import numpy as np
    from tests.helpers import load_golden
    from pointcloud_bridge_amd import ops, rowmlp
    g = load_golden("model_ptv3")
    xyz, colors = torch.from_numpy(g["xyz"]).cuda(), torch.from_numpy(g["colors"]).cuda()
    ref = g["logits_eval"]
    scale = np.abs(ref).max()
    model = _ptv3(g)
    try:
        rowmlp.set_precision("fp32")
        with torch.no_grad():
            got = model(xyz, colors).cpu().numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() / scale < 1e-4
        calls = []
        real = ops._launch
        monkeypatch.setattr(ops, "_launch", lambda name, *a, **k: (calls.append(name), real(name, *a, **k))[1])
        rowmlp.set_precision("bf16")
        with torch.no_grad():
            got16 = model(xyz, colors).float().cpu().numpy()
        assert calls.count("pcb_attention_fwd_bf16") == int(g["depth"])
        err = np.abs(got16 - ref).max() / scale
        assert err < 2e-2, err                                  # measured 5.8e-3 (fp32 mode: 6.6e-7)
        assert np.abs(got16 - ref).mean() / np.abs(ref).mean() < 1e-2   # measured 3.4e-3
        assert (got16.argmax(-1) == ref.argmax(-1)).mean() > 0.99      # measured 0.9992
        # gradients wanted: the reference's composition on ATen (no native backward in this row)
        out = model(xyz, colors)
        out.sum().backward()
        assert model.blocks[0].attn.qkv.weight.grad is not None
    finally:
        rowmlp.set_precision("fp32")


@pytest.mark.parametrize("R,C", [(1000, 384), (77, 96), (4096, 768), (33, 1024)])
@pytest.mark.parametrize("with_h,with_pos", [(False, True), (True, False), (True, True), (False, False)])
def test_add_layernorm_rows(R, C, with_h, with_pos):
    """pcb_add_layernorm_bf16 (csrc/tokens.hip) against nn.LayerNorm in fp32 on the same bf16 rows:
    `x = x + attn(norm1(x) + pos)` / `x = x + mlp(norm2(x))` of PointTransformerBlock.forward (:140-147) as one pass each."""
    from pointcloud_bridge_amd import ops
    g = torch.Generator().manual_seed(R + C)
    x = (torch.randn(R, C, generator=g) * 2 + 0.5).cuda().to(torch.bfloat16)
    h = torch.randn(R, C, generator=g).cuda().to(torch.bfloat16) if with_h else None
    pos = torch.randn(R, C, generator=g).cuda().to(torch.bfloat16) if with_pos else None
    norm = torch.nn.LayerNorm(C, eps=1e-6).cuda()
    with torch.no_grad():
        norm.weight.copy_(torch.rand(C, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(C, generator=g) * 0.2)
    xs, out = ops.add_layernorm(x, h, pos, norm, want_sum=True)
    x_ref = x.float() if h is None else (x.float() + h.float()).to(torch.bfloat16).float()
    if h is None:
        assert xs is None
    else:
        assert torch.equal(xs.float(), x_ref)                                   # the new residual stream, one rounding
    ref = torch.nn.functional.layer_norm(x_ref, (C,), norm.weight, norm.bias, 1e-6)
    if pos is not None:
        ref = ref + pos.float()
    err = (out.float() - ref).abs()
    assert float(err.max()) <= 2 ** -8 * float(ref.abs().max()) + 1e-6          # one bf16 rounding of the result
    assert float(err.mean()) <= 2 ** -9 * float(ref.abs().mean())


@pytest.mark.parametrize("R,H", [(1000, 1536), (31, 64), (4096, 384)])
def test_geglu_rows(R, H):
    """pcb_geglu_bf16 against x * F.gelu(gate) in fp32 (models/PointTransformerV3.py:8-21)."""
    from pointcloud_bridge_amd import ops
    g = torch.Generator().manual_seed(R + H)
    y = (torch.randn(R, 2 * H, generator=g) * 1.5).cuda().to(torch.bfloat16)
    out = ops.geglu(y)
    a, gate = y.float().chunk(2, dim=-1)
    ref = a * torch.nn.functional.gelu(gate)
    err = (out.float() - ref).abs()
    assert out.shape == (R, H)
    assert float((err / (ref.abs() + 1e-3)).max()) <= 2 ** -8 * 1.01 + 1e-6


def test_ptv3_fused_rows_equal_the_aten_rows(monkeypatch):
    """The fused row path of the bf16 inference pass (two add+LayerNorm launches and one GEGLU launch per block, cached
    bf16 weights) against the same pass on ATen row ops: logits within bf16 distance, same predictions."""
    from tests.helpers import load_golden
    from pointcloud_bridge_amd import rowmlp
    from pointcloud_bridge_amd.models import PointTransformerV3 as mod
    g = load_golden("model_ptv3")
    xyz, colors = torch.from_numpy(g["xyz"]).cuda(), torch.from_numpy(g["colors"]).cuda()
    model = _ptv3(g)
    with rowmlp.precision("bf16"), torch.no_grad():
        monkeypatch.setattr(mod, "_ROWS_FUSED", True)
        fused = model(xyz, colors).float()
        again = model(xyz, colors).float()          # (cached bf16 weights on the second call)
        monkeypatch.setattr(mod, "_ROWS_FUSED", False)
        aten = model(xyz, colors).float()
    assert torch.equal(fused, again)
    scale = float(aten.abs().max())
    assert float((fused - aten).abs().max()) < 2e-2 * scale
    assert float((fused.argmax(-1) == aten.argmax(-1)).float().mean()) > 0.99
