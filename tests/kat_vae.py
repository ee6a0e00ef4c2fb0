"""Known-answer vectors for the three diffusers classes that are NOT under /root/reference (CogVideoXDownsample3D,
CogVideoXUpsample3D, DiagonalGaussianDistribution), derived INDEPENDENTLY of oracle/vae_ref.py: integer-valued inputs, hand-chosen
kernels (identity / all-ones), expected outputs written out with explicit index loops in numpy from the published semantics
(diffusers 0.31: downsample = temporal average pooling that keeps the first frame when T is odd, zero pad right/bottom by one,
3x3 stride-2 convolution per frame; upsample = nearest x2 in space — and in time for compress_time, where an odd T > 1 keeps the first
frame single — followed by a 3x3 pad-1 convolution per frame).  Exact in fp32 and in bf16 (small integers)."""
import numpy as np


def video(T, H, W, C=1):
    """x[c, t, h, w] = 1 + c + 2 t + h + w: small integers, even in t so that pair averages are integers, and every 3x3 window sum stays
    below 256 (T <= 9, H = 4, W = 6: at most 9 * 25 = 225) — exact in bf16 as well as in fp32."""
    c, t, h, w = np.meshgrid(np.arange(C), np.arange(T), np.arange(H), np.arange(W), indexing="ij")
    return (1 + c + 2 * t + h + w).astype(np.float32)


def upsample_time_map(T, compress_time):
    """Output frame -> input frame."""
    if not compress_time or T == 1:
        return list(range(T))
    if T % 2 == 1:
        return [0] + [t for t in range(1, T) for _ in (0, 1)]
    return [t for t in range(T) for _ in (0, 1)]


def upsample_expected(x, compress_time):
    """Identity 3x3 kernel (centre tap 1, bias 0): the convolution returns the nearest-resized tensor itself."""
    C, T, H, W = x.shape
    tm = upsample_time_map(T, compress_time)
    out = np.zeros((C, len(tm), 2 * H, 2 * W), np.float32)
    for to, ti in enumerate(tm):
        for h in range(2 * H):
            for w in range(2 * W):
                out[:, to, h, w] = x[:, ti, h // 2, w // 2]
    return out


def pooled_frames(x, compress_time):
    C, T, H, W = x.shape
    if not compress_time:
        return x
    if T % 2 == 1:
        frames = [x[:, 0]] + [(x[:, t] + x[:, t + 1]) / 2 for t in range(1, T, 2)]
    else:
        frames = [(x[:, t] + x[:, t + 1]) / 2 for t in range(0, T, 2)]
    return np.stack(frames, axis=1)


def downsample_expected(x, compress_time):
    """All-ones 3x3 kernel over a single channel, bias 0: every output is the plain sum of its (zero padded right/bottom) 3x3 window."""
    p = pooled_frames(x, compress_time)
    C, T, H, W = p.shape
    assert C == 1
    pad = np.zeros((C, T, H + 1, W + 1), np.float32)
    pad[:, :, :H, :W] = p
    Ho, Wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
    out = np.zeros((1, T, Ho, Wo), np.float32)
    for t in range(T):
        for h in range(Ho):
            for w in range(Wo):
                out[0, t, h, w] = pad[0, t, 2 * h:2 * h + 3, 2 * w:2 * w + 3].sum()
    return out


CASES_T = (1, 2, 3, 9)

# DiagonalGaussianDistribution: parameters = [mean | logvar] along channels; logvar clamped to [-30, 20]; sample = mean + exp(logvar/2) * noise
GAUSS = dict(mean=[0.5, -1.0, 2.0, 0.0], logvar=[float(np.log(4.0)), 100.0, -100.0, 0.0], noise=[1.0, 1.0, 1.0, -3.0],
             sample=[0.5 + 2.0, -1.0 + float(np.exp(10.0)), 2.0 + float(np.exp(-15.0)), -3.0])
