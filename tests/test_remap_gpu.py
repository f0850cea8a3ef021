"""GPU parity of the stereo rectification (cv::remap INTER_LINEAR with float maps) against the CPU oracle: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rectification_like_maps(w, h, dw, dh, seed):
    """Smooth maps of the kind cv::initUndistortRectifyMap produces: a small rotation plus radial distortion."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:dh, 0:dw].astype(np.float64)
    cx, cy, f = w / 2 + rng.uniform(-5, 5), h / 2 + rng.uniform(-5, 5), 0.6 * w
    x, y = (xx - dw / 2) / f, (yy - dh / 2) / f
    th = rng.uniform(-0.02, 0.02)
    xr, yr = x * np.cos(th) - y * np.sin(th), x * np.sin(th) + y * np.cos(th)
    r2 = xr * xr + yr * yr
    k = 1 + rng.uniform(-0.3, 0.1) * r2 + rng.uniform(-0.05, 0.1) * r2 * r2
    return (f * xr * k + cx).astype(np.float32), (f * yr * k + cy).astype(np.float32)


@pytest.mark.parametrize("w,h,dw,dh", [(752, 480, 752, 480), (640, 480, 641, 479), (97, 61, 130, 90), (8, 8, 3, 2)])
def test_remap_rectification_maps(vsa, oracle, w, h, dw, dh):
    img = vsa.synth_frame(3, max(w, 64), max(h, 64))[:h, :w]
    mx, my = rectification_like_maps(w, h, dw, dh, w + dh)
    rm = vsa.Remapper(mx, my, w, h, device=0)
    got = rm.remap(img)
    want = oracle.remap_linear_u8(img, mx, my)
    assert np.array_equal(got, want), np.count_nonzero(got != want)
    rm.close()


def test_remap_hostile_maps(vsa, oracle):
    """Random positions far outside, exactly on the last row / column, NaN, infinities and values no int holds."""
    w, h = 200, 120
    img = vsa.synth_frame(9, w, h)
    rng = np.random.default_rng(2)
    mx = rng.uniform(-40, w + 40, (77, 211)).astype(np.float32); my = rng.uniform(-40, h + 40, (77, 211)).astype(np.float32)
    special = np.array([np.nan, np.inf, -np.inf, 3e9, -3e9, 1e30, w - 1, w - 1 + 1 / 64, w - 0.5, -1, -1 + 1 / 32, -0.5, -1 / 64, 0, 1023.99, 40000.0],
                       np.float32)
    mx[0, :len(special)] = special; my[0, :len(special)] = 5.25
    my[1, :len(special)] = special; mx[1, :len(special)] = 7.75
    mx[2, :len(special)] = special; my[2, :len(special)] = special[::-1]
    rm = vsa.Remapper(mx, my, w, h, device=0)
    assert np.array_equal(rm.remap(img), oracle.remap_linear_u8(img, mx, my))
    rm.close()


def test_remap_then_extract_device_resident(vsa, oracle):
    """The stereo front end with frames kept on the device: raw pair -> remap -> extraction, against the oracle's
    remap followed by the oracle's extraction."""
    import torch
    w, h, B = 752, 480, 4
    raw = np.stack([vsa.synth_frame(20 + i, w, h) for i in range(B)])
    mx, my = rectification_like_maps(w, h, w, h, 5)
    rm = vsa.Remapper(mx, my, w, h, device=0)
    p = vsa.make_params(1200)
    ex = vsa.Extractor(p, w, h, max_batch=B, device=0)
    st = torch.cuda.current_stream().cuda_stream
    d_raw = torch.from_numpy(raw).cuda(); d_rect = torch.empty_like(d_raw)
    cap = ex.cap
    d_k = torch.zeros(B * cap * 7, dtype=torch.float32, device="cuda"); d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(B, dtype=torch.int32, device="cuda")
    rm.remap_batch_device(d_raw.data_ptr(), w, w * h, B, d_rect.data_ptr(), w, w * h, st)
    ex.extract_batch_device(d_rect.data_ptr(), w, w * h, B, d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), cap, st)
    torch.cuda.synchronize()
    rect = d_rect.cpu().numpy(); c = d_c.cpu().numpy(); desc = d_d.cpu().numpy().reshape(B, cap, 32)
    oex = oracle.Extractor(oracle.make_params(1200))
    for f in range(B):
        want = oracle.remap_linear_u8(raw[f], mx, my)
        assert np.array_equal(rect[f], want)
        ok, od = oex.extract(want)
        assert c[f] == len(ok) and np.array_equal(desc[f, :c[f]], od)
    ex.close(); rm.close()
