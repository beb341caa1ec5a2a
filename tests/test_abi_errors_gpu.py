"""GPU: error conventions of the C ABI (int return codes + cap_last_error, no exceptions across the boundary)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(N, a, **over):
    from embodied_captioning_amd.engine import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD
    cfg = N.CapConfig()
    cfg.struct_size = C.sizeof(N.CapConfig)
    cfg.arch, cfg.compute_dtype = 0, N.CAP_F32
    cfg.image_size, cfg.patch_size = a.image_size, a.patch_size
    cfg.v_hidden, cfg.v_layers, cfg.v_heads, cfg.v_mlp, cfg.v_eps = a.v_hidden, a.v_layers, a.v_heads, a.v_mlp, a.v_eps
    cfg.t_hidden, cfg.t_layers, cfg.t_heads, cfg.t_ffn = a.t_hidden, a.t_layers, a.t_heads, a.t_ffn
    cfg.vocab, cfg.max_pos, cfg.t_eps = a.vocab, a.max_pos, a.t_eps
    cfg.bos, cfg.eos, cfg.pad = a.bos, a.eos, a.pad
    cfg.max_batch, cfg.max_beams, cfg.max_len = 4, 2, 12
    for i in range(3):
        cfg.pix_mean[i], cfg.pix_std[i] = OPENAI_CLIP_MEAN[i], OPENAI_CLIP_STD[i]
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def test_create_rejects_bad_configs():
    from embodied_captioning_amd import _native as N
    from embodied_captioning_amd.config import BlipArch
    lib = N.load_library()
    a = BlipArch.tiny()
    h = C.c_void_p()
    for over, word in [({"struct_size": 8}, "size mismatch"), ({"arch": 7}, "unknown arch"), ({"compute_dtype": 5}, "dtype"),
                       ({"v_heads": a.v_heads + 1}, "head_dim"), ({"max_len": a.max_pos + 1}, "capacity"),
                       ({"max_beams": 9}, "capacity"), ({"patch_size": 7}, "geometry")]:
        rc = lib.cap_create(C.byref(_cfg(N, a, **over)), C.byref(h))
        assert rc != 0 and word in N.last_error(), (over, N.last_error())
    assert lib.cap_create(None, C.byref(h)) != 0


def test_calls_fail_with_codes_not_crashes():
    from embodied_captioning_amd import _native as N
    from embodied_captioning_amd.config import BlipArch
    from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels
    lib = N.load_library()
    a = BlipArch.tiny()
    h = C.c_void_p()
    assert lib.cap_create(C.byref(_cfg(N, a)), C.byref(h)) == 0
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    px = synthetic_pixels(4, a.image_size, seed=0).cuda()
    ids = torch.zeros((4, 12), dtype=torch.int32, device="cuda")
    args = lambda B, K, L: (h, C.c_void_p(px.data_ptr()), N.CAP_PIX_F32_NCHW, B, K, L, C.c_float(1.0), C.c_void_p(ids.data_ptr()),  # noqa: E731
                            None, None, None, s)
    # weights not loaded yet: the count of missing tensors comes back and generate refuses
    missing = lib.cap_finalize_weights(h)
    assert missing > 50 and "not loaded" in N.last_error()
    assert lib.cap_generate(*args(4, 1, 12)) != 0
    # a tensor with the wrong shape is rejected, an unknown name is reported as "not mine" (1), not as an error
    t = torch.zeros(3, 5)
    shape = (C.c_int64 * 2)(3, 5)
    assert lib.cap_load_weight(h, b"vision_model.post_layernorm.weight", C.c_void_p(t.data_ptr()), 0, 2, shape, s) < 0
    assert "elements" in N.last_error()
    assert lib.cap_load_weight(h, b"some.other.tensor", C.c_void_p(t.data_ptr()), 0, 2, shape, s) == 1
    for name, w in procedural_blip_state_dict(a, 0).items():
        w = w.contiguous()
        shp = (C.c_int64 * max(w.dim(), 1))(*(w.shape if w.dim() else (1,)))
        assert lib.cap_load_weight(h, name.encode(), C.c_void_p(w.data_ptr()), 0, max(w.dim(), 1), shp, s) >= 0
    assert lib.cap_finalize_weights(h) == 0
    assert lib.cap_generate(*args(4, 1, 12)) == 0
    for B, K, L, word in [(5, 1, 12, "capacity"), (4, 3, 12, "capacity"), (4, 1, 13, "capacity"), (0, 1, 12, "capacity")]:
        assert lib.cap_generate(*args(B, K, L)) != 0 and word in N.last_error(), (B, K, L, N.last_error())
    assert lib.cap_generate(h, None, N.CAP_PIX_F32_NCHW, 4, 1, 12, C.c_float(1.0), C.c_void_p(ids.data_ptr()), None, None, None, s) != 0
    assert lib.cap_generate(h, C.c_void_p(px.data_ptr()), 9, 4, 1, 12, C.c_float(1.0), C.c_void_p(ids.data_ptr()), None, None, None, s) != 0
    assert lib.cap_embed_text(h, C.c_void_p(ids.data_ptr()), C.c_void_p(ids.data_ptr()), 1, 4, C.c_void_p(ids.data_ptr()), s) != 0
    assert "sentence encoder" in N.last_error()
    assert lib.cap_generate(None, C.c_void_p(px.data_ptr()), 0, 4, 1, 12, C.c_float(1.0), C.c_void_p(ids.data_ptr()), None, None, None, s) != 0
    torch.cuda.synchronize()
    assert lib.cap_destroy(h) == 0 and lib.cap_destroy(None) == 0


def test_single_kernel_entry_points_validate_shapes():
    from embodied_captioning_amd import _native as N
    lib = N.load_library()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    A = torch.zeros(64, 100, device="cuda", dtype=torch.bfloat16)          # K = 100 is not a whole number of 64-wide slabs
    W = torch.zeros(64, 100, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    assert lib.cap_op_gemm(1, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), None, None, C.c_void_p(out.data_ptr()), 64, 64,
                           100, 0, 0, 0, s) != 0 and "multiple" in N.last_error()
    assert lib.cap_op_vit_attention(1, C.c_void_p(A.data_ptr()), C.c_void_p(out.data_ptr()), 1, 130, 1, 2, s) != 0
    assert "key blocks" in N.last_error()
    x = torch.zeros(4, 30, device="cuda")
    assert lib.cap_op_layernorm(0, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), C.c_float(1e-5),
                                C.c_void_p(x.data_ptr()), None, 4, 30, s) != 0
