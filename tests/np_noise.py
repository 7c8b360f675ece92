"""Vectorised numpy fp32 restatement of the 2-D noise bases, written from SURVEY.md Appendix A in the
float4 form Unity.Mathematics uses (test infrastructure; a second formulation of what oracle/noize_oracle.c
states scalar by scalar).  Every operation is an IEEE binary32 ufunc: no contraction, source order."""
import numpy as np

f = np.float32


def mod289(x): return x - np.floor(x * (f(1.0) / f(289.0))) * f(289.0)
def mod7(x): return x - np.floor(x * (f(1.0) / f(7.0))) * f(7.0)
def permute(x): return mod289((f(34.0) * x + f(1.0)) * x)
def taylor_inv_sqrt(r): return f(1.79284291400159) - f(0.85373472095314) * r
def fade(t): return t * t * t * (t * (t * f(6.0) - f(15.0)) + f(10.0))
def frac(x): return x - np.floor(x)
def lerp(a, b, s): return a + s * (b - a)


def cnoise2(px, py):  # A.2
    px, py = np.asarray(px, f), np.asarray(py, f)
    Pi = [np.floor(px) + f(0.0), np.floor(py) + f(0.0), np.floor(px) + f(1.0), np.floor(py) + f(1.0)]
    Pf = [frac(px) - f(0.0), frac(py) - f(0.0), frac(px) - f(1.0), frac(py) - f(1.0)]
    Pi = [mod289(v) for v in Pi]
    ix = [Pi[0], Pi[2], Pi[0], Pi[2]]
    iy = [Pi[1], Pi[1], Pi[3], Pi[3]]
    fx = [Pf[0], Pf[2], Pf[0], Pf[2]]
    fy = [Pf[1], Pf[1], Pf[3], Pf[3]]
    gx, gy = [], []
    for k in range(4):
        i = permute(permute(ix[k]) + iy[k])
        g = frac(i * (f(1.0) / f(41.0))) * f(2.0) - f(1.0)
        gyk = np.abs(g) - f(0.5)
        tx = np.floor(g + f(0.5))
        gx.append(g - tx)
        gy.append(gyk)
    # g00 = (gx.x, gy.x), g10 = (gx.y, gy.y), g01 = (gx.z, gy.z), g11 = (gx.w, gy.w)
    # norm = taylorInvSqrt((g00.g00, g01.g01, g10.g10, g11.g11))
    n00 = taylor_inv_sqrt(gx[0] * gx[0] + gy[0] * gy[0])
    n01 = taylor_inv_sqrt(gx[2] * gx[2] + gy[2] * gy[2])
    n10 = taylor_inv_sqrt(gx[1] * gx[1] + gy[1] * gy[1])
    n11 = taylor_inv_sqrt(gx[3] * gx[3] + gy[3] * gy[3])
    g00 = (gx[0] * n00, gy[0] * n00)
    g01 = (gx[2] * n01, gy[2] * n01)
    g10 = (gx[1] * n10, gy[1] * n10)
    g11 = (gx[3] * n11, gy[3] * n11)
    d00 = g00[0] * fx[0] + g00[1] * fy[0]
    d10 = g10[0] * fx[1] + g10[1] * fy[1]
    d01 = g01[0] * fx[2] + g01[1] * fy[2]
    d11 = g11[0] * fx[3] + g11[1] * fy[3]
    fdx, fdy = fade(Pf[0]), fade(Pf[1])
    nx0, nx1 = lerp(d00, d10, fdx), lerp(d01, d11, fdx)
    return f(2.3) * lerp(nx0, nx1, fdy)


def snoise2(vx, vy, permute_mul=34.0, invert_i1=False, cy_scale=1.0):  # A.3
    """The three keyword arguments are MUTATIONS for negative controls (tests/test_reference_screenshots.py): a hash
    multiplier other than 34, the simplex corner select inverted, the skew constant C.y scaled.  Defaults = the spec."""
    vx, vy = np.asarray(vx, f), np.asarray(vy, f)
    Cx, Cy, Cz, Cw = f(0.211324865405187), f(0.366025403784439), f(-0.577350269189626), f(0.024390243902439)
    Cy = f(Cy * f(cy_scale))

    def permute(x):  # shadows the module's permute only when mutated
        return mod289((f(permute_mul) * x + f(1.0)) * x)
    s = vx * Cy + vy * Cy                      # dot(v, C.yy)
    ix, iy = np.floor(vx + s), np.floor(vy + s)
    t = ix * Cx + iy * Cx                      # dot(i, C.xx)
    x0x, x0y = vx - ix + t, vy - iy + t
    gt = (x0x > x0y) != bool(invert_i1)
    i1x, i1y = np.where(gt, f(1.0), f(0.0)), np.where(gt, f(0.0), f(1.0))
    x12 = [x0x + Cx, x0y + Cx, x0x + Cz, x0y + Cz]
    x12[0] = x12[0] - i1x
    x12[1] = x12[1] - i1y
    ix, iy = mod289(ix), mod289(iy)
    p = [permute(permute(iy + f(0.0)) + ix + f(0.0)), permute(permute(iy + i1y) + ix + i1x),
         permute(permute(iy + f(1.0)) + ix + f(1.0))]
    m = [np.maximum(f(0.5) - (x0x * x0x + x0y * x0y), f(0.0)),
         np.maximum(f(0.5) - (x12[0] * x12[0] + x12[1] * x12[1]), f(0.0)),
         np.maximum(f(0.5) - (x12[2] * x12[2] + x12[3] * x12[3]), f(0.0))]
    m = [v * v for v in m]
    m = [v * v for v in m]
    x = [f(2.0) * frac(v * Cw) - f(1.0) for v in p]
    h = [np.abs(v) - f(0.5) for v in x]
    ox = [np.floor(v + f(0.5)) for v in x]
    a0 = [a - b for a, b in zip(x, ox)]
    m = [mk * (f(1.79284291400159) - f(0.85373472095314) * (a * a + hh * hh)) for mk, a, hh in zip(m, a0, h)]
    g0 = a0[0] * x0x + h[0] * x0y
    g1 = a0[1] * x12[0] + h[1] * x12[1]
    g2 = a0[2] * x12[2] + h[2] * x12[3]
    return f(130.0) * (m[0] * g0 + m[1] * g1 + m[2] * g2)


def rgrad2(px, py, rot):  # A.1
    u = permute(permute(px) + py) * f(0.0243902439) + rot
    u = frac(u) * f(6.28318530718)
    return np.cos(u.astype(np.float64)).astype(f), np.sin(u.astype(np.float64)).astype(f), u


def psrnoise2_parts(posx, posy, perx=1010.0, pery=102.0):  # A.4 up to the gradient hash arguments
    posx, posy = np.asarray(posx, f), np.asarray(posy, f)
    posy = posy + f(0.001)
    uvx, uvy = posx + posy * f(0.5), posy
    i0x, i0y = np.floor(uvx), np.floor(uvy)
    f0x, f0y = frac(uvx), frac(uvy)
    gt = f0x > f0y
    i1x, i1y = np.where(gt, f(1.0), f(0.0)), np.where(gt, f(0.0), f(1.0))
    p0 = (i0x - i0y * f(0.5), i0y)
    p1 = (p0[0] + i1x - i1y * f(0.5), p0[1] + i1y)
    p2 = (p0[0] + f(0.5), p0[1] + f(1.0))
    xw = [np.fmod(p[0], f(perx)) for p in (p0, p1, p2)]
    yw = [np.fmod(p[1], f(pery)) for p in (p0, p1, p2)]
    iu = [a + f(0.5) * b for a, b in zip(xw, yw)]
    d = [(posx - p[0], posy - p[1]) for p in (p0, p1, p2)]
    return iu, yw, d


def cellular2(Px, Py, permute_mul=34.0, jitter=1.0):  # A.5
    """The keyword arguments are MUTATIONS for negative controls (tests/test_reference_screenshots.py).  Defaults = the spec."""
    Px, Py = np.asarray(Px, f), np.asarray(Py, f)
    K, Ko, jitter = f(0.142857142857), f(0.428571428571), f(jitter)

    def permute(x):  # shadows the module's permute only when mutated
        return mod289((f(permute_mul) * x + f(1.0)) * x)
    Pix, Piy = mod289(np.floor(Px)), mod289(np.floor(Py))
    Pfx, Pfy = frac(Px), frac(Py)
    oi = [f(-1.0), f(0.0), f(1.0)]
    of = [f(-0.5), f(0.5), f(1.5)]
    px = [permute(Pix + o) for o in oi]
    cols = []
    for c, xoff in enumerate((f(0.5), f(-0.5), f(-1.5))):
        dcol = []
        for k in range(3):
            p = permute(px[c] + Piy + oi[k])
            ox = frac(p * K) - Ko
            oy = mod7(np.floor(p * K)) * K - Ko
            dx = Pfx + xoff + jitter * ox
            dy = Pfy - of[k] + jitter * oy
            dcol.append(dx * dx + dy * dy)
        cols.append(dcol)
    d1, d2, d3 = cols
    d1a = [np.minimum(a, b) for a, b in zip(d1, d2)]
    d2 = [np.maximum(a, b) for a, b in zip(d1, d2)]
    d2 = [np.minimum(a, b) for a, b in zip(d2, d3)]
    d1 = [np.minimum(a, b) for a, b in zip(d1a, d2)]
    d2 = [np.maximum(a, b) for a, b in zip(d1a, d2)]
    lt = d1[0] < d1[1]
    d1[0], d1[1] = np.where(lt, d1[0], d1[1]), np.where(lt, d1[1], d1[0])
    lt = d1[0] < d1[2]
    d1[0], d1[2] = np.where(lt, d1[0], d1[2]), np.where(lt, d1[2], d1[0])
    d1[1], d1[2] = np.minimum(d1[1], d2[1]), np.minimum(d1[2], d2[2])
    d1[1] = np.minimum(d1[1], d1[2])
    d1[1] = np.minimum(d1[1], d2[0])
    return np.sqrt(d1[0]), np.sqrt(d1[1])
