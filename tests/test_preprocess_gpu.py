"""GPU: cap_crop_resize_u8 against Pillow itself (crop + BICUBIC resize), bit-exact."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _pil(frame_rgb, r, S):
    return np.asarray(Image.fromarray(frame_rgb).crop(tuple(int(v) for v in r)).resize((S, S), resample=Image.BICUBIC))


def test_device_tables_equal_host_tables_bitwise():
    """cap_crop_resize_tables (fp64 on the device, contraction off) against the host's numpy tables, which
    tests/test_preprocess_cpu.py holds to the restatement that Pillow pins: every crop size 1..600 and a few large ones."""
    import ctypes as C
    from embodied_captioning_amd import _native as N
    from embodied_captioning_amd.preprocess import pil_bicubic_coeffs, shorter_side_geometry
    lib = N.load_library()
    S = 224
    sizes = list(range(1, 601)) + [640, 719, 1080, 1919, 4000]
    n = len(sizes)
    rects = np.array([[3, 5, 3 + w, 5 + sizes[(i * 7) % n]] for i, w in enumerate(sizes)], dtype=np.int32)
    for center in (False, True):
        geom = np.array([shorter_side_geometry(int(r[2] - r[0]), int(r[3] - r[1]), S) if center else (S, S, 0, 0) for r in rects],
                        dtype=np.int32)
        sc = lambda a, b: np.maximum(a / b, 1.0)                                                     # noqa: E731
        KH = int((np.ceil(2 * sc((rects[:, 2] - rects[:, 0]).astype(float), geom[:, 0].astype(float))) * 2 + 1).max())
        KV = int((np.ceil(2 * sc((rects[:, 3] - rects[:, 1]).astype(float), geom[:, 1].astype(float))) * 2 + 1).max())
        rd, gd = torch.from_numpy(rects).cuda(), torch.from_numpy(geom).cuda()
        hb = torch.full((n, S, 2), -1, dtype=torch.int32, device="cuda"); vb = torch.full_like(hb, -1)
        hk = torch.full((n, S, KH), -1, dtype=torch.int32, device="cuda")
        vk = torch.full((n, S, KV), -1, dtype=torch.int32, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())                                                      # noqa: E731
        rc = lib.cap_crop_resize_tables(p(rd), p(gd), n, S, KH, KV, p(hb), p(hk), p(vb), p(vk),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.cap_last_error()
        torch.cuda.synchronize()
        hb, hk, vb, vk = (t.cpu().numpy() for t in (hb, hk, vb, vk))
        for i, r in enumerate(rects):
            for size_in, tot, off, b_d, k_d in ((r[2] - r[0], geom[i, 0], geom[i, 2], hb[i], hk[i]),
                                                (r[3] - r[1], geom[i, 1], geom[i, 3], vb[i], vk[i])):
                b_h, k_h = pil_bicubic_coeffs(int(size_in), int(tot))
                assert np.array_equal(b_d, b_h[off:off + S]), (center, r)
                assert np.array_equal(k_d[:, :k_h.shape[1]], k_h[off:off + S]) and not k_d[:, k_h.shape[1]:].any(), (center, r)


@pytest.mark.parametrize("tables", ["device", "host"])
@pytest.mark.parametrize("S", [224, 384])
def test_crops_equal_pillow_bitwise(S, tables):
    from embodied_captioning_amd.preprocess import crop_resize_u8 as _cr
    crop_resize_u8 = lambda *a, **k: _cr(*a, tables=tables, **k)                                    # noqa: E731
    rng = np.random.default_rng(S)
    H, W = 480, 640
    frame = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    rects = [(0, 0, W, H), (10, 20, 234, 244), (100, 50, 130, 470), (600, 400, 640, 480), (5, 5, 7, 8), (0, 0, 1, 1),
             (17, 300, 630, 333), (200, 100, 424, 324), (320, 0, 321, 480),
             (500, 400, 700, 520), (-20, -10, 90, 60), (0, 0, 480, 640)]       # leaving the frame: Image.crop pads with zeros
    out = crop_resize_u8(frame, rects, S).cpu().numpy()
    for i, r in enumerate(rects):
        assert np.array_equal(out[i], _pil(frame, r, S)), r


def test_bgr_swap_and_device_frame():
    """The reference swaps BGR -> RGB before cropping (pseudolabeler.py:670): same result from a BGR device tensor."""
    from embodied_captioning_amd.preprocess import crop_resize_u8
    rng = np.random.default_rng(1)
    frame_bgr = rng.integers(0, 256, size=(256, 300, 3), dtype=np.uint8)
    rgb = np.ascontiguousarray(frame_bgr[:, :, ::-1])
    rects = [(3, 4, 250, 200), (100, 100, 160, 130)]
    out = crop_resize_u8(torch.from_numpy(frame_bgr).cuda(), rects, 224, bgr=True).cpu().numpy()
    for i, r in enumerate(rects):
        assert np.array_equal(out[i], _pil(rgb, r, 224))


def test_rejects_empty_rectangles():
    from embodied_captioning_amd.preprocess import crop_resize_u8
    frame = np.zeros((64, 64, 3), dtype=np.uint8)
    for bad in [(5, 5, 5, 9), (9, 5, 3, 9), (0, 0, 10, 0)]:
        with pytest.raises(ValueError):
            crop_resize_u8(frame, [bad], 224)
    assert crop_resize_u8(frame, [], 224).shape == (0, 224, 224, 3)


def test_shorter_side_resize_and_centre_crop_equals_pillow():
    """CoCa's transform (aspect-preserving bicubic resize of the shorter side, centre crop) on the device: the kept window
    of the resized crop, byte for byte."""
    from embodied_captioning_amd.preprocess import crop_resize_u8, shorter_side_geometry
    rng = np.random.default_rng(7)
    frame = rng.integers(0, 256, size=(300, 500, 3), dtype=np.uint8)
    rects = [(0, 0, 500, 300), (10, 10, 110, 290), (200, 100, 460, 180), (50, 60, 274, 284), (3, 3, 40, 30)]
    S = 224
    out = crop_resize_u8(frame, rects, S, center_crop=True).cpu().numpy()
    for i, r in enumerate(rects):
        im = Image.fromarray(frame).crop(r)
        nw, nh, left, top = shorter_side_geometry(im.size[0], im.size[1], S)
        ref = np.asarray(im.resize((nw, nh), resample=Image.BICUBIC).crop((left, top, left + S, top + S)))
        assert np.array_equal(out[i], ref), r


@pytest.mark.parametrize("arch_name,model_name", [("blip", "procedural-blip-tiny:1"), ("blip2", "procedural-blip2-tiny:1"), ("coca", "procedural-coca-tiny:1")])
def test_wrappers_preprocess_pil_lists_on_the_device_bit_exact(arch_name, model_name):
    """`generate_batch` / `forward` receive PIL crops: the processor's bicubic resize (BLIP / BLIP-2: square; CoCa: shorter side + centre
    crop) runs on the device by default (`captioner.device_resize`) and returns the bytes host Pillow returns - odd sizes, up- and
    down-scaling, non-RGB modes."""
    import numpy as np
    import torch
    from PIL import Image
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    rng = np.random.default_rng(11)
    ims = []
    for (w, h), mode in (((37, 91), "RGB"), ((400, 33), "RGB"), ((224, 224), "RGB"), ((5, 7), "RGB"), ((130, 64), "L"), ((61, 200), "RGBA"), ((640, 480), "RGB")):
        ch = {"RGB": 3, "L": 1, "RGBA": 4}[mode]
        a = rng.integers(0, 256, size=(h, w, ch) if ch > 1 else (h, w), dtype=np.uint8)
        ims.append(Image.fromarray(a, mode))
    kw = dict(arch_name=arch_name, model_name=model_name, height=224, width=224, dtype="f32")
    dev = select_captioner(Configuration(**kw).captioner).eval()
    host = select_captioner(Configuration(device_resize=False, **kw).captioner).eval()
    assert dev.device_resize and not host.device_resize
    a, b = dev.preprocess(ims), host.preprocess(ims)
    assert a.is_cuda and a.dtype == torch.uint8 and tuple(a.shape) == tuple(b.shape)
    assert torch.equal(a.cpu(), b)
    one = dev.preprocess(ims[0])
    assert torch.equal(one.cpu(), b[:1])
    assert dev(ims[1])["text"] == host(ims[1])["text"]


@pytest.mark.parametrize("center_crop", [False, True])
def test_boxes_of_several_frames_in_one_call_bit_exact(center_crop):
    """`crop_resize_u8_frames`: boxes of several frames of different sizes (only the in-frame pixels travel, packed), BGR frames,
    boxes that leave their frame partly or wholly - every crop equal to Pillow's crop + bicubic resize (or shorter side + centre crop)."""
    import numpy as np
    import torch
    from PIL import Image
    from embodied_captioning_amd.preprocess import crop_resize_u8_frames, shorter_side_geometry
    rng = np.random.default_rng(4)
    S = 48
    frames = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in ((120, 160), (64, 64), (300, 200), (37, 91))]
    rects = [[(10, 20, 90, 100), (-15, 30, 60, 140), (100, -8, 175, 50)], [(0, 0, 64, 64)], [], [(5, 3, 30, 36), (200, 200, 230, 260), (-40, -40, 20, 20)]]
    out = crop_resize_u8_frames(frames, rects, S, bgr=True, center_crop=center_crop).cpu().numpy()
    k = 0
    for fr, rs in zip(frames, rects):
        pil = Image.fromarray(np.ascontiguousarray(fr[:, :, ::-1]))
        for r in rs:
            c = pil.crop(r)
            if center_crop:
                nw, nh, left, top = shorter_side_geometry(c.size[0], c.size[1], S)
                want = np.asarray(c.resize((nw, nh), resample=Image.BICUBIC).crop((left, top, left + S, top + S)))
            else:
                want = np.asarray(c.resize((S, S), resample=Image.BICUBIC))
            assert np.array_equal(out[k], want), (k, r)
            k += 1
    assert k == out.shape[0] == 7
    assert crop_resize_u8_frames(frames[:1], [[]], S).shape == (0, S, S, 3)


def test_long_lists_go_through_in_byte_bounded_groups(monkeypatch):
    """The pinned staging buffer of `resize_u8_list` / `crop_resize_u8_frames` is bounded: a list beyond the bound is resized group by
    group - same bytes as in one go."""
    import numpy as np
    import torch
    from embodied_captioning_amd import preprocess as P
    rng = np.random.default_rng(8)
    ims = [rng.integers(0, 256, size=(20 + i % 13, 31 + i % 7, 3), dtype=np.uint8) for i in range(40)]
    whole = P.resize_u8_list(ims, 32)
    monkeypatch.setattr(P, "_STAGE_BYTES", 9000)                 # a few images per group
    assert len(P._byte_groups([a.shape[:2] for a in ims])) > 5
    assert torch.equal(P.resize_u8_list(ims, 32), whole)
