"""Known-answer and cross-implementation tests of the CPU oracle.

The reference ships no tests, so these are authored: analytic answers, hand-computed small cases,
and an independent numpy (fp32, same operation order) restatement of each stencil that the C
oracle must equal bit for bit.
"""
import numpy as np
import pytest

from conftest import adversarial_tiles

f32 = np.float32


# ---- independent numpy restatements ---------------------------------------------------------------
def np_pass_x(a, k, factor):
    o = (len(k) - 1) // 2
    p = np.pad(a, ((0, 0), (o, o)), mode="edge")
    tot = np.zeros_like(a)
    for kk in range(len(k)):  # k ascending, KernelOperators.cs:34-40
        tot = tot + p[:, kk:kk + a.shape[1]] * f32(k[kk])
    return tot * f32(factor)


def np_pass_z(a, k, factor):
    o = (len(k) - 1) // 2
    p = np.pad(a, ((o, o), (0, 0)), mode="edge")
    tot = np.zeros_like(a)
    for kk in range(len(k)):  # k = o - kk descending, Kernel[k_off - k] = k[kk], KernelOperators.cs:59-65
        off = 2 * o - kk
        tot = tot + p[off:off + a.shape[0], :] * f32(k[kk])
    return tot * f32(factor)


def np_erosion(a):
    px = np.pad(a, ((0, 0), (1, 0)), mode="edge")
    a = np.minimum(px[:, :-1], px[:, 1:])
    pz = np.pad(a, ((1, 0), (0, 0)), mode="edge")
    return np.minimum(pz[:-1, :], pz[1:, :])


def sh(a, dz, dx):
    """a sampled at (z+dz, x+dx) with clamp-to-edge."""
    p = np.pad(a, 1, mode="edge")
    return p[1 + dz:1 + dz + a.shape[0], 1 + dx:1 + dx + a.shape[1]]


def np_flow_step(h, w, fN, fS, fE, fW):
    tot = w + h
    dW, dE, dS, dN = tot - sh(tot, 0, -1), tot - sh(tot, 0, 1), tot - sh(tot, -1, 0), tot - sh(tot, 1, 0)
    z = f32(0)
    flW, flE, flS, flN = (np.maximum(z, fW + dW), np.maximum(z, fE + dE), np.maximum(z, fS + dS),
                          np.maximum(z, fN + dN))
    s = (flW + flE) + (flS + flN)   # math.csum(float4): pairwise
    with np.errstate(divide="ignore", invalid="ignore"):
        K = np.clip(w / (s * f32(0.2)), f32(0), f32(1))
    pos = s > 0
    return tuple(np.where(pos, f * K, z).astype(f32) for f in (flN, flS, flE, flW))


def np_water_step(w, fN, fS, fE, fW):
    out = ((fW + fE) + fS) + fN
    inn = f32(0) + sh(fE, 0, -1)
    inn = inn + sh(fW, 0, 1)
    inn = inn + sh(fN, -1, 0)
    inn = inn + sh(fS, 1, 0)
    return np.maximum(f32(0), w + (inn - out) * f32(0.2))


def np_velocity(fN, fS, fE, fW):
    dl, dr = sh(fE, 0, -1) - fW, fE - sh(fW, 0, 1)
    dt, db = sh(fS, 1, 0) - fN, fS - sh(fN, -1, 0)
    vx, vy = (dl + dr) * f32(0.5), (dt + db) * f32(0.5)
    return np.sqrt(vx * vx + vy * vy)


# ---- fractal --------------------------------------------------------------------------------------
def test_b1_normalisation_ignores_starting_amplitude(oracle):
    a1 = oracle.fractal(oracle.SIMPLEX, 16, 16, 0.5, 1.0, 2.0, 0.0, 4, 3, 5, 37)
    a2 = oracle.fractal(oracle.SIMPLEX, 16, 16, 0.5, 2.0, 2.0, 0.0, 4, 3, 5, 37)
    assert np.array_equal(a2, a1 * f32(2))  # power-of-two scaling is exact
    assert oracle.fractal_norm(0.5, 4, 1.0) == oracle.fractal_norm(0.5, 4, 7.0)
    G = f32(np.exp2(f32(-0.5)))
    t, a = f32(0), f32(1)
    for _ in range(4):
        t = f32(t + a)
        a = f32(a * G)
    assert oracle.fractal_norm(0.5, 4) == t


def test_b2_detune_applies_before_first_frequency_step(oracle):
    hurst, amp, step, det, ns = 0.3, 1.5, 2.0, 0.03, 50
    x, z, xp, zp = 7, 11, 100, 200
    xi, zi = f32(f32(x) + f32(xp)) / f32(ns), f32(f32(z) + f32(zp)) / f32(ns)
    G = f32(np.exp2(f32(-hurst)))
    f1 = f32(f32(step) - f32(det))
    n0 = f32(oracle.noise_value(oracle.PERLIN, float(xi), float(zi)))
    n1 = f32(oracle.noise_value(oracle.PERLIN, float(f32(f1 * xi)), float(f32(f1 * zi))))
    t = f32(f32(0) + f32(f32(amp) * n0))
    t = f32(t + f32(f32(f32(amp) * G) * n1))
    want = f32(t / oracle.fractal_norm(hurst, 2))
    got = oracle.fractal_cell(oracle.PERLIN, x, z, hurst, amp, step, det, 2, xp, zp, ns)
    assert got == want


def test_b3_world_offsets_are_added_cells(oracle):
    R = 24
    for basis in (oracle.SIMPLEX, oracle.PERLIN, oracle.CELLULAR, oracle.ROTATED_SIMPLEX):
        mono = oracle.fractal(basis, R, 2 * R, 0.4, 1.0, 2.0, 0.0, 5, 1000, 2000, 170)
        right = oracle.fractal(basis, R, R, 0.4, 1.0, 2.0, 0.0, 5, 1000 + R, 2000, 170)
        below = oracle.fractal(basis, R, R, 0.4, 1.0, 2.0, 0.0, 5, 1000, 2000 + 7, 170)
        assert np.array_equal(mono[:, R:], right)
        assert np.array_equal(mono[7:, :R], below[:R - 7])


def test_b4_perlin_is_half_on_the_integer_lattice(oracle):
    a = oracle.fractal(oracle.PERLIN, 8, 8, 0.0, 1.0, 2.0, 0.0, 1, 0, 0, 1)  # xi = x exactly
    assert np.array_equal(a, np.full((8, 8), 0.5, f32))
    assert oracle.cnoise2(5.0, 9.0) == 0.0


def test_noise_ranges_and_continuity(oracle):
    rng = np.random.default_rng(7)
    pts = (rng.random((4000, 2)) * 60).astype(f32)
    for fn, lo, hi in ((oracle.snoise2, -1.0, 1.0), (oracle.cnoise2, -1.0, 1.0),
                       (lambda x, y: oracle.psrnoise2(x, y, rot=0.62), -1.01, 1.01)):
        v = np.array([fn(float(x), float(y)) for x, y in pts])
        assert lo <= v.min() < -0.8 and 0.8 < v.max() <= hi
    for basis in range(8):
        v = np.array([oracle.noise_value(basis, float(x), float(y)) for x, y in pts[:1000]])
        assert v.min() >= -0.01 and v.max() <= 1.4, basis  # rectified bases live in ~[0,1]
    xs = np.arange(0, 12, 1e-3, dtype=f32)
    for basis in (oracle.SIMPLEX, oracle.PERLIN, oracle.ROTATED_SIMPLEX, oracle.CELLULAR, oracle.DR_PERLIN,
                  oracle.DR_SIMPLEX):
        v = np.array([oracle.noise_value(basis, float(x), float(f32(0.37) * x + f32(0.11))) for x in xs])
        assert np.abs(np.diff(v)).max() < 0.01, basis  # no jumps across lattice cells


def test_psrnoise_hash_is_a_small_integer(oracle):
    # the HIP kernel tabulates rgrad2 by this value (nz_fractal.hip); it must be an integer in [0, 289]
    # arguments: iu = xw + 0.5 yw in (-1061, 1061), yw in (-102, 102); C fmod keeps the dividend's sign, so
    # negative lattice coordinates hash negative arguments
    hs = [oracle.psr_hash(float(x), float(y)) for x in np.arange(-1062.0, 1062.0, 0.5)
          for y in (-101.0, -57.0, -1.0, 0.0, 1.0, 57.0, 101.0)]
    assert min(hs) >= 0.0 and max(hs) <= 289.0 and all(float(h).is_integer() for h in hs)


def test_psrnoise_is_periodic_in_y(oracle):
    for x, y in ((3.25, 7.5), (100.125, 40.75), (555.5, 99.0)):
        a, b = oracle.psrnoise2(x, y), oracle.psrnoise2(x, y + 102.0)
        assert abs(a - b) < 2e-3


# ---- separable filters ----------------------------------------------------------------------------
@pytest.mark.parametrize("ft", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13])
def test_kernel_filter_equals_numpy_restatement(oracle, ft):
    kx, kz, factor, ks = oracle.kernel_filter_table(ft)
    for name, t in adversarial_tiles(33).items():
        want = np_pass_z(np_pass_x(t, kx, factor), kz, factor)
        got = oracle.kernel_filter(t, ft)
        assert np.array_equal(got, want), (ft, name)


def test_b6_pass_order_x_ascending_z_descending(oracle):
    ones = np.ones(3, f32)
    row = np.zeros((3, 3), f32)
    row[1] = [1e8, -1e8, 1.0]
    assert oracle.pass_sample_x(row, 3, ones, 1.0)[1, 1] == 1.0        # ((1e8 + -1e8) + 1)
    assert oracle.pass_sample_z(row.T.copy(), 3, ones, 1.0)[1, 1] == 0.0  # ((1 + -1e8) + 1e8)


def test_b7_impulse_response_is_the_outer_product(oracle):
    R, c = 17, 8
    imp = np.zeros((R, R), f32)
    imp[c, c] = 1.0
    k = oracle.kernel_filter_table(oracle.GAUSS5_S1)[0]
    got = oracle.kernel_filter(imp, oracle.GAUSS5_S1)
    assert np.array_equal(got[c - 2:c + 3, c - 2:c + 3], np.outer(k, k).astype(f32))
    assert got.sum() == pytest.approx(1.0, abs=1e-6)
    # asymmetric bodies: the X pass correlates, the Z pass indexes Kernel[k_off - k] (flipped)
    got = oracle.kernel_filter(imp, oracle.SOBEL3_H)   # kx = {-1,0,1}, kz = {1,2,1}
    assert list(got[c, c - 1:c + 2]) == [2.0, 0.0, -2.0] and list(got[c - 1, c - 1:c + 2]) == [1.0, 0.0, -1.0]
    got = oracle.kernel_filter(imp, oracle.PREWITT3_V)  # kx = {1,1,1}, kz = {-1,0,1}
    assert list(got[c - 1:c + 2, c]) == [-1.0, 0.0, 1.0]


def test_b5_b8_clamp_to_edge_keeps_constant_tiles_constant(oracle):
    t = np.full((20, 20), 0.5, f32)
    for ft in (oracle.GAUSS5_S1, oracle.GAUSS9_S2, oracle.SMOOTH3):
        out = oracle.kernel_filter(t, ft, iterations=3)
        assert np.all(out == out[0, 0]) and abs(out[0, 0] - 0.5) < 1e-6
    # an impulse in the corner keeps more mass than one in the middle (edge taps fold back)
    imp = np.zeros((9, 9), f32)
    imp[0, 0] = 1.0
    k = oracle.kernel_filter_table(oracle.GAUSS5_S1)[0]
    out = oracle.kernel_filter(imp, oracle.GAUSS5_S1)
    assert out[0, 0] == f32(f32(f32(k[0] + k[1]) + k[2]) * f32(f32(k[0] + k[1]) + k[2]))


def test_gauss_and_smooth_entry_points(oracle):
    t = adversarial_tiles(40)["uniform"]
    for sigma in (0, 3, 15):
        for width in (3, 5, 9, 25):
            k = oracle.gauss_kernel(sigma, width)
            assert np.array_equal(oracle.gauss(t, width, sigma), np_pass_z(np_pass_x(t, k, 1.0), k, 1.0))
    k = np.full(7, f32(1) / f32(7), f32)
    assert np.array_equal(oracle.smooth(t, 7), np_pass_z(np_pass_x(t, k, 1.0), k, 1.0))
    # GaussFilter.Schedule with an even width: 5-tap body, kernelSize 4 -> k_off 1 -> taps 0..2 (BlurJob.cs:11-21)
    k5 = oracle.gauss_kernel(3, 4)
    assert np.array_equal(oracle.gauss(t, 4, 3), np_pass_z(np_pass_x(t, k5[:3], 1.0), k5[:3], 1.0))


def test_b9_min_window_is_minus_one_zero(oracle):
    t = np.ones((7, 7), f32)
    t[3, 3] = 0.0
    out = oracle.erosion_min(t)
    want = np.ones((7, 7), f32)
    want[3:5, 3:5] = 0.0
    assert np.array_equal(out, want)
    for name, a in adversarial_tiles(21).items():
        assert np.array_equal(oracle.erosion_min(a), np_erosion(a)), name
        assert np.array_equal(oracle.erosion_min(a, 3), np_erosion(np_erosion(np_erosion(a)))), name


# ---- flow map -------------------------------------------------------------------------------------
def test_flow_steps_equal_numpy_restatement(oracle):
    rng = np.random.default_rng(5)
    R = 19
    h = rng.random((R, R), dtype=f32)
    w = (rng.random((R, R), dtype=f32) * f32(0.01)).astype(f32)
    fl = [(rng.random((R, R), dtype=f32) * f32(0.02)).astype(f32) for _ in range(4)]
    got = oracle.flow_step(h, w, *fl)
    want = np_flow_step(h, w, *fl)
    for g, wv, n in zip(got, want, "NSEW"):
        assert np.array_equal(g, wv), n
    assert np.array_equal(oracle.water_step(w, *got), np_water_step(w, *got))
    assert np.array_equal(oracle.velocity(*got), np_velocity(*got))


def test_flowmap_equals_composition(oracle):
    rng = np.random.default_rng(6)
    R = 23
    h = rng.random((R, R), dtype=f32)
    w = np.full((R, R), 0.0001, f32)
    fl = [np.zeros((R, R), f32) for _ in range(4)]
    for _ in range(4):
        fl = list(np_flow_step(h, w, *fl))
        w = np_water_step(w, *fl)
    v = np_velocity(*fl)
    want = (v - f32(0.0)) / f32(f32(0.005) - f32(0.0))
    assert np.array_equal(oracle.flowmap(h, 4, 0.0, 0.005), want)


def test_b12_flat_terrain_has_zero_velocity(oracle):
    out = oracle.flowmap(np.full((12, 12), 0.3, f32), 5, -0.1, 0.1)
    assert np.array_equal(out, np.full((12, 12), (f32(0) - f32(-0.1)) / f32(f32(0.1) - f32(-0.1)), f32))


def test_b13_b14_ramp_first_iteration(oracle):
    R = 8
    s = f32(1.0 / 64.0)
    h = np.tile(np.arange(R, dtype=f32) * s, (R, 1))
    w = np.full((R, R), 0.0001, f32)
    z = np.zeros((R, R), f32)
    fN, fS, fE, fW = oracle.flow_step(h, w, z, z, z, z)
    # water runs downhill to the west: flow = slope * K, K = w / (slope * 0.2)
    K = f32(f32(0.0001) / f32(s * f32(0.2)))
    assert np.all(fE == 0) and np.all(fN == 0) and np.all(fS == 0)
    assert np.all(fW[:, 0] == 0)  # clamped neighbour is the cell itself: no gradient
    assert np.allclose(fW[:, 1:], s * K, rtol=1e-6)
    w2 = oracle.water_step(w, fN, fS, fE, fW)
    assert np.allclose(w2[:, 1:R - 1], 0.0001, rtol=1e-5)       # in == out
    assert np.allclose(w2[:, R - 1], 0.0001, rtol=1e-5)         # border cell receives its own outflow (B14)
    assert np.allclose(w2[:, 0], 0.0001 + float(s * K) * 0.2, rtol=1e-5)


def test_b15_normalise_divides_even_for_an_empty_range(oracle):
    a = np.full((4, 4), 0.7, f32)
    with np.errstate(all="ignore"):
        out = oracle.normalize(a, 0.2, 0.2)
    assert np.all(np.isneginf(out))  # v forced to 0, (0 - 0.2) / 0
    assert np.array_equal(oracle.normalize(a, 0.0, 0.5), (a - f32(0)) / f32(0.5))


# ---- mesh -----------------------------------------------------------------------------------------
def test_b18_index_buffer_closed_form(oracle):
    R = 3
    heights = np.zeros((R + 4, R + 4), f32)
    _, idx = oracle.mesh_heightmap(oracle.MESH_OVERSHOOT, heights, R, 2, 10.0, 30.0)
    want = []
    for z in range(1, R + 1):
        for x in range(1, R + 1):
            vi = (R + 1) * z + x
            want += [vi - R - 2, vi - 1, vi - R - 1, vi - R - 1, vi - 1, vi]
    assert idx.tolist() == want
    assert idx.max() == (R + 1) ** 2 - 1


def test_b16_mesh_vertices_by_hand(oracle):
    R, IR, H, TS = 2, 6, 10.0, 30.0
    hts = (np.arange(IR * IR, dtype=f32).reshape(IR, IR) / f32(IR * IR)).astype(f32)
    vtx, _ = oracle.mesh_heightmap(oracle.MESH_OVERSHOOT, hts, R, 2, H, TS)
    off = 2
    assert vtx.shape == (9, 12)
    # x = 0 special case and the regular positions
    assert vtx[0, 0] == -(f32(0.5) * f32(TS) / f32(R)) and vtx[1, 0] == f32(1) * f32(TS) / f32(R) - f32(0.5)
    assert vtx[3, 2] == f32(1) * f32(TS) / f32(R) - f32(0.5)
    z, x = 1, 2
    v = vtx[z * (R + 1) + x]
    t = hts[z + off, x + off]
    l, r, u, d = hts[z + off, x - 1 + off], hts[z + off, x + 1 + off], hts[z - 1 + off, x + off], hts[z + 1 + off, x + off]
    assert v[1] == t * f32(H)
    n = np.array([(l - r) / f32(2) * f32(8), f32(2) / f32(H), (u - d) / f32(2) * f32(8)], f32)
    n = (f32(1) / np.sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])) * n
    assert np.array_equal(v[3:6], n)
    assert v[6] == (u - d) / f32(2) * f32(0) - f32(4) * ((r - l) / f32(2)) and v[7] == 16.0 and v[9] == 0.0
    assert v[8] == f32(0) * ((r - l) / f32(2)) - (u - d) / f32(2) * f32(4)
    assert v[10] == f32(x) / (f32(R) - f32(0.5)) and v[11] == f32(z) / (f32(R) - f32(0.5))
    # SquareGrid: edge-extrapolated neighbours, uv / (R + 1)
    vs, _ = oracle.mesh_heightmap(oracle.MESH_SQUARE, hts, R, 2, H, TS)
    v0 = vs[0]
    t = hts[off, off]
    l = t - (hts[off, 1 + off] - t)          # InterpolateEdge(t, h(x+1))
    r = hts[off, 1 + off]                    # x < R - 1
    u = hts[1 + off, off] - (t - hts[1 + off, off])  # InterpolateEdge(h(z+1), t)
    d = hts[1 + off, off]
    n = np.array([(l - r) / f32(2) * f32(8), f32(2) / f32(H), (u - d) / f32(2) * f32(8)], f32)
    n = (f32(1) / np.sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])) * n
    assert np.array_equal(v0[3:6], n)
    assert vs[4, 10] == f32(1) / (f32(R) + f32(1))


def test_b17_mesh_rejects_margins_that_index_outside_the_plane(oracle):
    hts = np.zeros((8, 8), f32)
    with pytest.raises(ValueError):
        oracle.mesh_heightmap(oracle.MESH_OVERSHOOT, hts, 8, 0, 1.0, 1.0)   # off = 0, reads column 8
    with pytest.raises(ValueError):
        oracle.mesh_heightmap(oracle.MESH_OVERSHOOT, hts, 6, 1, 1.0, 1.0)   # off = 1, reads row 8
    with pytest.raises(ValueError):
        oracle.mesh_heightmap(oracle.MESH_SQUARE, hts, 8, 0, 1.0, 1.0)
    oracle.mesh_heightmap(oracle.MESH_OVERSHOOT, hts, 4, 2, 1.0, 1.0)
    oracle.mesh_heightmap(oracle.MESH_SQUARE, hts, 7, 0, 1.0, 1.0)


def test_pipeline_equals_stage_composition(oracle):
    R = 48
    a = oracle.fractal(oracle.SIMPLEX, R, R, 0.4, 1.0, 2.0, 0.0, 13, 0, 0, 1700)
    a = oracle.kernel_filter(a, oracle.GAUSS5_S1, 3)
    a = oracle.flowmap(a, 2, 0.0, 0.005)
    a = oracle.erosion_min(a, 2)
    b = oracle.pipeline(R, R, gauss_iterations=3, flow_iterations=2, erosion_iterations=2)
    assert np.array_equal(a, b)


# ---- element-wise stages ------------------------------------------------------
def test_constant_reduce_curve_known_answers(oracle):
    a = np.array([[0.1, 0.5], [0.75, 1.5]], f32)
    b = np.array([[0.4, 0.5], [0.25, -2.0]], f32)
    assert np.array_equal(oracle.constant(a, oracle.CONST_MULTIPLY, 0.5), a * f32(0.5))
    assert np.array_equal(oracle.constant(a, oracle.CONST_BINARIZE, 0.5), np.array([[0, 1], [1, 1]], f32))  # >=
    assert np.array_equal(oracle.reduce(a, b, oracle.RED_SUBTRACT), a - b)
    assert np.array_equal(oracle.reduce(a, b, oracle.RED_MULTIPLY), a * b)
    assert np.array_equal(oracle.reduce(a, b, oracle.RED_ROOTSUMSQUARES), np.sqrt(a * a + b * b))
    assert np.array_equal(oracle.reduce(a, b, oracle.RED_MAX), np.maximum(a, b))
    assert np.array_equal(oracle.reduce(a, b, oracle.RED_MIN), np.minimum(a, b))
    # CurveOperator.Apply: rect = clamp(v)*N, lower = min(floor(rect), N-2), lerp, clamp to [0,1]
    n = 4
    samples = np.array([0.0, 0.2, 0.9, 1.2], f32)
    v = np.array([[-1.0, 0.0, 0.3], [0.5, 0.99, 7.0]], f32)
    got = oracle.curve(v, samples)
    want = []
    for x in v.reshape(-1):
        rect = f32(min(max(x, f32(0)), f32(1)) * f32(n))
        lo = min(np.floor(rect), f32(n - 2))
        val = samples[int(lo)] + f32(rect - lo) * f32(samples[int(lo) + 1] - samples[int(lo)])
        want.append(min(f32(1), max(f32(0), f32(val))))
    assert np.array_equal(got.reshape(-1), np.array(want, f32))
    assert got[1, 2] == 1.0  # v = 1 -> rect = 4 -> lower 2, t = 2 -> extrapolates past the last sample, clamped


def test_thermal_erosion_known_answers(oracle):
    # ThermalErosionFilter.cs: maxDiff = tan(talus) * ratio / res; pairs steeper than maxDiff move `increment` of
    # the excess each way; four phases of disjoint 2x2 blocks starting at (x, z) = (1|2, 2|1)
    res = 8
    a = np.zeros((res, res), f32)
    a[2, 1] = 1.0
    out = oracle.thermal_erosion(a, 45.0, 0.5, 0.75, 1)
    md = f32(np.tan(f32(f32(45.0 / 90.0) * f32(3.14159)) / f32(2.0)) * f32(0.75)) / f32(res)
    assert out.sum() == pytest.approx(1.0, abs=1e-6)      # mass is conserved
    assert out[2, 1] < 1.0 and out[2, 2] > 0 and out[3, 1] > 0
    assert np.array_equal(out[0], a[0]) and np.array_equal(out[:, 0], a[:, 0])  # row 0 / column 0 are never touched
    flat = np.full((res, res), 0.25, f32)
    assert np.array_equal(oracle.thermal_erosion(flat, 30.0, 0.5, 0.75, 3), flat)
    # a slope gentler than the talus angle is a fixed point
    ramp = np.tile(np.arange(res, dtype=f32) * (md * f32(0.5)), (res, 1))
    assert np.array_equal(oracle.thermal_erosion(ramp, 45.0, 0.5, 0.75, 2), ramp)


def test_crop_takes_the_top_left_corner_and_clamps(oracle):
    # Filter/Sample/CropJob.cs:43-59 never sets Offset: the "centre crop" is the top-left corner; reads clamp
    a = np.arange(36, dtype=f32).reshape(6, 6)
    assert np.array_equal(oracle.crop(a, 4), a[:4, :4])
    big = oracle.crop(a, 8)
    assert np.array_equal(big[:6, :6], a) and np.array_equal(big[7], big[5]) and np.array_equal(big[:, 7], big[:, 5])


# ---- second formulation of the 2-D noise bases (tests/np_noise.py, float4 form of SURVEY.md Appendix A) ----
@pytest.mark.parametrize("scale", [20.0, 3000.0, 60000.0, 3.0e6])
def test_noise_bases_equal_the_vectorised_numpy_restatement(oracle, scale):
    import np_noise as N
    rng = np.random.default_rng(int(scale))
    n = 3000
    x = ((rng.random(n, dtype=f32) - f32(0.5)) * f32(scale)).astype(f32)
    y = ((rng.random(n, dtype=f32) - f32(0.5)) * f32(scale)).astype(f32)
    x[:8] = [-8959.0, -8959.5, -17629.25, 8959.0, 0.0, -0.0, 289.0, -289.0]   # lattice cells where fp32 mod289 is 289
    y[:8] = [-18207.0, 0.5, -8959.0, -8959.75, 0.0, -1.0, -289.0, 578.0]
    pts = list(zip(x.tolist(), y.tolist()))
    assert np.array_equal(np.array([oracle.cnoise2(u, v) for u, v in pts], f32), N.cnoise2(x, y))
    assert np.array_equal(np.array([oracle.snoise2(u, v) for u, v in pts], f32), N.snoise2(x, y))
    c = np.array([oracle.cellular2(u, v) for u, v in pts], f32)
    F1, F2 = N.cellular2(x, y)
    assert np.array_equal(c[:, 0], F1) and np.array_equal(c[:, 1], F2)
    # psrnoise: the gradient hash (bit-exact) and the value (cos/sin from a different libm: 1e-6)
    iu, yw, d = N.psrnoise2_parts(x, y)
    for k in range(3):
        h = np.array([oracle.psr_hash(a, b) for a, b in zip(iu[k].tolist(), yw[k].tolist())], f32)
        assert np.array_equal(h, N.permute(N.permute(iu[k]) + yw[k]))
    for rot in (0.0, 0.62):
        w, t4 = [], []
        for k in range(3):
            gx, gy, _ = N.rgrad2(iu[k], yw[k], f32(rot))
            w.append(gx * d[k][0] + gy * d[k][1])
            t = np.maximum(f32(0.8) - (d[k][0] * d[k][0] + d[k][1] * d[k][1]), f32(0.0))
            t2 = t * t
            t4.append(t2 * t2)
        want = f32(11.0) * (t4[0] * w[0] + t4[1] * w[1] + t4[2] * w[2])
        got = np.array([oracle.psrnoise2(u, v, 1010.0, 102.0, rot) for u, v in pts], f32)
        assert np.allclose(got, want, rtol=0, atol=2e-6)


def test_square_planar_mesh_by_hand(oracle):
    # SharedSquareGridPosition.cs:20-50 at resolution 2: 9 vertices, 8 triangles
    vtx, idx = oracle.mesh_square_grid(2)
    assert vtx.shape == (9, 12) and idx.shape == (24,)
    assert np.array_equal(vtx[:, 3:10], np.tile(np.array([0, 0, -1, 1, 0, 0, -1], f32), (9, 1)))  # normal, tangent
    assert np.array_equal(vtx[:, 0].reshape(3, 3), np.tile(np.array([-0.5, 0.0, 0.5], f32), (3, 1)))
    assert np.array_equal(vtx[:, 2].reshape(3, 3), np.tile(np.array([[-0.5], [0.0], [0.5]], f32), (1, 3)))
    assert np.array_equal(vtx[:, 1], np.zeros(9, f32))
    third = f32(1.0) / f32(3.0)
    assert np.array_equal(vtx[:, 10].reshape(3, 3)[0], np.array([0.0, third, f32(2.0) / f32(3.0)], f32))
    assert np.array_equal(idx[:6], np.array([0, 3, 1, 1, 3, 4], np.uint32))   # vi=4: (4-4, 4-1, 4-3), (4-3, 4-1, 4)


# ---- live erosion grid jobs (planes indexed [x, z]) ---------------------------------------------------------------
def test_update_flow_from_track_known_answers(oracle):
    # LiveErosionDataTypes.cs:869-886
    pool = np.array([[0.0, 0.001], [0.00004, 0.0]], f32)
    flow = np.array([[0.5, 0.5], [0.5, 0.0]], f32)
    track = np.array([[0.0, 3.0], [0.02, 0.0]], f32)
    p, fl, tr = oracle.update_flow_from_track(pool, flow, track, 0.05, 0.1, 1000.0)
    assert np.array_equal(tr, np.zeros((2, 2), f32))
    assert fl[0, 0] == f32(f32(1.0) - f32(0.05)) * f32(0.5)                       # no pool, no track: decay
    assert fl[0, 1] == f32(f32(1.0) - f32(0.1) * f32(0.05)) * f32(0.5)              # pool: slow decay, track ignored
    want = f32(f32(f32(1.0) - f32(0.05)) * f32(0.5)) + f32(f32(f32(0.05) * f32(50.0)) * f32(0.02)) / f32(f32(1.0) + f32(f32(50.0) * f32(0.02)))
    assert fl[1, 0] == f32(want)                                                   # track feeds the flow
    evap = f32(0.1) / f32(1000.0)
    assert p[0, 1] == f32(0.001) - evap and p[1, 0] == 0.0 and p[0, 0] == 0.0      # evaporation, floored at 0


def test_pool_automata_known_answers(oracle):
    # MultiThreadErosionJob.cs:264-327 + WorldTile.SpreadPool (LiveErosionDataTypes.cs:938-1010)
    res = 8
    # (1) two wet cells in a walled basin exchange a quarter of their level difference per visit:
    #     A = (3,3) is visited in colour pass (xoff 0, zoff 1), B = (4,3) in pass (1, 1)
    walls = np.full((res, res), 10.0, f32)
    walls[3, 3] = walls[4, 3] = 0.0
    pool = np.zeros((res, res), f32)
    pool[3, 3], pool[4, 3] = 1.0, 0.2
    out = oracle.pool_automata(pool, walls, 1)
    assert out[3, 3] == pytest.approx(0.7, abs=1e-6) and out[4, 3] == pytest.approx(0.5, abs=1e-6)
    assert (out > 0).sum() == 2
    level = oracle.pool_automata(pool, walls, 60)
    assert abs(level[3, 3] - level[4, 3]) < 1e-3 and level.sum() == pytest.approx(1.2, abs=1e-5)
    # (2) next to dry land that is not higher, the whole pool "drains" into the first such neighbour: on flat
    #     land a lone pool hops around as a unit, on a slope it runs downhill
    flat = np.zeros((res, res), f32)
    pool = np.zeros((res, res), f32)
    pool[4, 4] = 1.0
    one = oracle.pool_automata(pool, flat, 1)
    assert one.sum() == 1.0 and (one > 0).sum() == 1
    slope = np.tile(np.arange(res, dtype=f32)[:, None], (1, res))           # height grows with x
    pool = np.zeros((res, res), f32)
    pool[5, 3] = 0.5
    out = oracle.pool_automata(pool, slope, 1)
    assert out[5, 3] == 0.0 and out[:5, :].sum() == pytest.approx(0.5)     # everything ran to lower x
    assert not oracle.pool_automata(np.zeros((res, res), f32), slope, 3).any()
    # (3) less than 1e-3 of water never moves
    pool = np.zeros((res, res), f32)
    pool[2, 2] = 5e-4
    assert np.array_equal(oracle.pool_automata(pool, flat, 4), pool)


def test_get_map_range_known_answers(oracle):
    # GetMapRangeJob (Filter/NormalizeJob.cs:33-43): sequential fold with math.min / math.max from the two limits
    f = np.float32
    r = oracle.get_map_range(np.array([3, -2, 7, 0.5], f))
    assert r.tolist() == [-2.0, 7.0, 9.0]
    r = oracle.get_map_range(np.array([3, np.nan, 7], f))                      # NaN cells are skipped
    assert r.tolist() == [3.0, 7.0, 4.0]
    r = oracle.get_map_range(np.array([3, 4], f), lim_min=1.0, lim_max=10.0)   # the limits take part in the fold
    assert r.tolist() == [1.0, 10.0, 9.0]
    r = oracle.get_map_range(np.array([np.nan, np.nan], f))                    # nothing but NaN: the limits stay
    assert r[0] == np.inf and r[1] == -np.inf
    # on a tie the later operand stays: among the zeros of a plane whose minimum is zero, the LAST one gives the sign
    r = oracle.get_map_range(np.array([0.0, 5, -0.0, 2], f))
    assert r[0] == 0 and np.signbit(r[0]) and r[1] == 5
    r = oracle.get_map_range(np.array([-0.0, 5, 0.0, 2], f))
    assert r[0] == 0 and not np.signbit(r[0])
    r = oracle.get_map_range(np.array([-3, 0.0, -0.0, -1], f))                 # ... and of a maximum that is zero
    assert r[1] == 0 and np.signbit(r[1]) and r[0] == -3


# ---- Unity.Mathematics' clamp on a NaN (round 6: the soak's seed 62067) -----------------------------------------------------
def test_flow_scale_of_zero_water_over_a_flux_sum_that_underflows_is_one(oracle):
    # ComputeFlowStep (FlowMapComponents.cs:52-56): K = clamp(water_0 / (sum_ * TIMESTEP), 0, 1) under `if (sum_ > 0)`.  With no
    # water on the cell and a flux sum so small that sum_ * 0.2 underflows to 0 -- a denormal height difference -- K is 0 / 0 = NaN,
    # and math.clamp = max(0, min(1, K)) with Unity.Mathematics' min / max (float.IsNaN(y) || x < y ? x : y: a NaN second operand
    # is skipped) gives 1: the fluxes stay what they were, nothing turns NaN.
    tiny = np.float32(1e-45)                      # the smallest denormal
    h = np.full((3, 3), tiny, f32)
    h[1, 2] = 0.0                                 # [z][x]: the cell EAST of the centre is one denormal lower
    water = np.zeros((3, 3), f32)
    z = np.zeros((3, 3), f32)
    fN, fS, fE, fW = oracle.flow_step(h, water, z, z, z, z)
    for fl in (fN, fS, fE, fW):
        assert np.isfinite(fl).all()
    # centre cell: flow = (0, 1e-45, 0, 0), sum_ = 1e-45 > 0, sum_ * 0.2 rounds to 0, K = 0 / 0 -> clamp -> 1
    assert fE[1, 1] == tiny and fW[1, 1] == 0 and fN[1, 1] == 0 and fS[1, 1] == 0
    # ... and the whole stage stays finite on such a plane
    big = np.zeros((40, 40), f32)
    big[7, 9] = tiny
    assert np.isfinite(oracle.flowmap(big, 3, 0.0, 0.005)).all()


def test_curve_reads_a_nan_cell_as_one(oracle):
    # CurveOperator.Apply (Filter/Curve/CurveJob.cs:56-89): rect = clamp(v, 0, 1) * N; clamp(NaN, 0, 1) = max(0, min(1, NaN)) = 1
    samples = np.linspace(0.0, 1.0, 5, dtype=f32) ** 2
    a = np.array([[np.nan, 1.0], [0.25, 2.0]], f32)
    got = oracle.curve(a, samples)
    assert got[0, 0] == got[0, 1] == got[1, 1] and np.isfinite(got).all()
