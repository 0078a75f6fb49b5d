"""Bit-reproducible fp32 exp / sigmoid used by the box pipeline (TEST INFRASTRUCTURE).

torchvision's detection head calls ``torch.sigmoid`` (score filter) and ``torch.exp``
(``BoxCoder.decode_single``).  Both are libm-style functions whose last bit differs between
CPU vector ISAs and GPUs, while NMS keep indices depend on the exact bits of the scores
(ordering) and of the decoded boxes (IoU).  To make keep indices reproducible across the CPU
checker and the HIP kernels, the pipeline fixes ONE op-by-op fp32 algorithm for ``exp`` built
only from IEEE-754 correctly rounded operations (mul, add, sub, div, round-to-nearest-even,
integer bit moves) with NO fused multiply-add.  This file is the numpy statement of that
algorithm; ``oracle/od_oracle.c`` and ``object_detection_amd/csrc/od_math.h`` state the same
sequence in C and HIP.  Accuracy: <= 2 ulp versus the exact value, i.e. ~2e-7 relative --
far inside the 1e-5 relative tolerance demanded for scores and decoded boxes.

Algorithm (Cephes-style expf, every operation a separately rounded fp32 op):
    k  = rint(x * LOG2E)                       # round half to even
    r  = (x - k * LN2_HI) - k * LN2_LO         # Cody-Waite, k*LN2_HI is exact
    p  = ((((((C6*r + C5)*r + C4)*r + C3)*r + C2)*r + C1) * (r*r) + r) + 1
    exp = (p * 2**k1) * 2**k2,  k1 = trunc(k/2), k2 = k - k1
with x > EXP_HI -> +inf and x < EXP_LO -> 0.
"""
import numpy as np

F = np.float32
LOG2E = F(1.44269504088896341)
LN2_HI = F(0.693359375)          # 0x3f318000: 9 significant bits, k*LN2_HI exact for |k| <= 2^15
LN2_LO = F(-2.12194440e-4)       # ln2 - LN2_HI
C6 = F(1.9875691500e-4)
C5 = F(1.3981999507e-3)
C4 = F(8.3334519073e-3)
C3 = F(4.1665795894e-2)
C2 = F(1.6666665459e-1)
C1 = F(5.0000001201e-1)
EXP_HI = F(88.72283935546875)    # largest x with finite fp32 exp
EXP_LO = F(-103.97208404541015625)  # below this exp underflows to 0 (smallest subnormal/2)


def _pow2(k):
    """2**k as fp32 for integer arrays k in [-126, 127] (exact, via exponent field)."""
    return ((k.astype(np.int32) + 127) << 23).astype(np.uint32).view(np.float32)


def od_expf(x):
    """Spec exp; x any array-like, computed in fp32, returns fp32 ndarray."""
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        xc = np.minimum(np.maximum(x, EXP_LO), EXP_HI)   # clamp (NaN handled below)
        k = np.rint(xc * LOG2E).astype(np.float32)
        r = (xc - k * LN2_HI) - k * LN2_LO
        p = C6 * r + C5
        p = p * r + C4
        p = p * r + C3
        p = p * r + C2
        p = p * r + C1
        p = p * (r * r) + r
        p = p + F(1.0)
        ki = k.astype(np.int32)
        k1 = (ki / 2).astype(np.int32)     # trunc toward zero
        k2 = ki - k1
        out = (p * _pow2(k1)) * _pow2(k2)
        out = np.where(x > EXP_HI, F(np.inf), out)
        out = np.where(x < EXP_LO, F(0.0), out)
        out = np.where(np.isnan(x), F(np.nan), out)
    return out.astype(np.float32)


def od_sigmoidf(x):
    """Spec sigmoid = 1 / (1 + od_expf(-x)), each op rounded to fp32."""
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(over="ignore", divide="ignore"):
        e = od_expf(-x)
        return (F(1.0) / (F(1.0) + e)).astype(np.float32)
