"""The oracle -- and the HIP path -- against the only OUTPUTS the reference holds: its README screenshots.

/root/reference/README.md:23-40 shows docs~/0.jpg ... 6.jpg: editor captures with the preview plane on the left and every
parameter readable in the inspector on the right.  tests/golden/make_screenshot_fixtures.py cropped the preview planes
(548^2 pixels = 1000^2 cells, uint8) into tests/golden/screenshots.npz and wrote the inspector values to screenshots.json.
This file computes the same planes at those parameters and compares them with the pictures.  It is an IMAGE-level pin
(JPEG, 8 bit, an editor display mapping), not the 1e-5 bar; it is what separates "internally consistent" from "matches
the reference's own output" for the recalled third-party noise (call sites
/root/reference/Noise/Fractal/Fractal.cs:234,269), the fBm loop (Fractal.cs:114-131), Gauss5 (Filter/Kernel/KernelJob.cs:99,
240-245; KernelOperators.cs:32-66) and the flow map (Geologic/FlowMap/FlowMapComponents.cs:20-104).

What was determined from the pictures (once, on 3.jpg; then held fixed for every other image and for every control):
  * orientation: image row = z (top = 0), image column = R-1-x  (the preview plane is mirrored in x);
  * display mapping: sRGB-decoded pixel value  ~  k * max(v - 0.5, 0)  (k differs per capture: 1.76 in 3.jpg, 0.20 in
    4.jpg; Pearson r does not see it).  Binning 3.jpg by the oracle's value gives exactly 0 below 0.5 and a straight line
    in linear light above;
  * scale: the 1096-pixel crop is 1000 cells (548 fixture pixels x 1.8248).  The only nuisance parameter left is the
    crop's shift (the frame detection is good to a few pixels): a translation within +-6 cells is fitted per comparison
    -- for the true model and for every negative control alike.
  * blue-tinted captures (2, 5, 6): R (= G) is the Gauss-filtered height under the same mapping, B is the flow map
    (monotone in the normalised velocity, saturating in its top 15 %).  JPEG chroma subsampling blurs B: the model is
    smoothed with sigma = 1 cell before comparing.

What the pictures pin, and what they do not (also DESIGN.md section 2):
  pinned  : snoise 2-D + fBm (r > 0.999), cellular 2-D (r > 0.99, hurst 1 as the inspector says, not the README's 0.4),
            Gauss5 x17 (r > 0.999; the filtered plane fits 4.jpg / 1.jpg better than the unfiltered one and vice versa),
            flow map x5 on the simplex terrain (r > 0.9; iterations 4-5 fit best), first flow iteration on the cellular
            terrain (r > 0.85).
  not     : cnoise, psrnoise, the 3-D bases, sin (no picture).  Min-erosion: 6.jpg's history cannot be reconstructed
            (its inspector reads Gauss 18 with the flow box last clicked; its R channel is the Gauss-filtered height at
            r > 0.998, its B channel matches none of the 40 orders of erosion / flow tried, r <= 0.17), so it stays
            pinned by the reference's code alone.  Flow iterations 2..5 on the cellular terrain: heights there differ by
            less than the water column between neighbours, so the picture depends on the screenshot-era component's
            water constants, which the inspector does not show (r falls from 0.88 at 1 iteration to 0.20 at 5).
"""
import json
import os
import sys

import numpy as np
import pytest
from scipy import ndimage, optimize

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import GOLDEN  # noqa: E402

f32 = np.float32
RES, PIX = 1000, 548
SCALE = RES / PIX
INSET = 6          # fixture pixels ignored along every edge (clamped samples of a shifted model)
SHIFT = 6          # cells


@pytest.fixture(scope="module")
def shots():
    z = np.load(os.path.join(GOLDEN, "screenshots.npz"))
    with open(os.path.join(GOLDEN, "screenshots.json")) as f:
        meta = json.load(f)["images"]
    return {k: srgb_to_linear(z[k]) for k in z.files}, meta


def srgb_to_linear(u8):
    c = np.asarray(u8, np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(np.float32)


def display(v):
    """What the preview shows of a plane of values: mirrored in x, black below 0.5, linear above."""
    return np.maximum(np.asarray(v, np.float32)[:, ::-1] - f32(0.5), f32(0.0))


def pearson(a, b):
    a = a.astype(np.float64).ravel()
    b = b.astype(np.float64).ravel()
    a -= a.mean()
    b -= b.mean()
    d = np.sqrt((a * a).sum() * (b * b).sum())
    return float((a * b).sum() / d) if d > 0 else 0.0


def _warp(plane, tx, ty, step=1):
    yy = SCALE * np.arange(0, PIX, step) + ty
    xx = SCALE * np.arange(0, PIX, step) + tx
    Y, X = np.meshgrid(yy, xx, indexing="ij")
    return ndimage.map_coordinates(plane, [Y, X], order=1, mode="nearest")


def fit(model, image):
    """(r, tx, ty): best Pearson r of `model` (a displayed 1000^2 plane, cells) against `image` (548^2, linear light)
    over a translation of at most SHIFT cells -- integer grid on every second pixel, then a simplex refinement on all."""
    m = ndimage.uniform_filter(np.asarray(model, np.float32), 2)   # the fixture's 2 x 2 box
    core = (slice(INSET, -INSET), slice(INSET, -INSET))
    img, img2 = image[core], image[::2, ::2][INSET // 2:-(INSET // 2), INSET // 2:-(INSET // 2)]

    def score(t):
        if max(abs(t[0]), abs(t[1])) > SHIFT + 0.5:
            return -1.0
        return pearson(_warp(m, t[0], t[1])[core], img)

    grid = [(pearson(_warp(m, tx, ty, 2)[INSET // 2:-(INSET // 2), INSET // 2:-(INSET // 2)], img2), tx, ty)
            for tx in range(-SHIFT, SHIFT + 1) for ty in range(-SHIFT, SHIFT + 1)]
    _, tx, ty = max(grid)
    res = optimize.minimize(lambda t: -score(t), [tx, ty], method="Nelder-Mead",
                            options=dict(xatol=0.02, fatol=1e-6, maxiter=60,
                                         initial_simplex=[[tx, ty], [tx + 0.7, ty], [tx, ty + 0.7]]))
    return -float(res.fun), float(res.x[0]), float(res.x[1])


def match(model, image):
    return fit(model, image)[0]


def _shifted(model, image):
    """`model` resampled onto the fixture grid at its best translation (for statistics other than r itself)."""
    _, tx, ty = fit(model, image)
    return _warp(ndimage.uniform_filter(np.asarray(model, np.float32), 2), tx, ty)


def highpass(a, sigma=6.0):
    return a - ndimage.gaussian_filter(a, sigma)


# ---- the models, from the oracle ------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def planes(oracle, shots):
    _, meta = shots
    s, c = meta["3"], meta["0"]
    assert (s["noiseType"], s["hurst"], s["octaves"], s["xpos"], s["zpos"], s["noiseSize"]) == ("Simplex", 0.422, 13, 0, 424, 1757)
    assert (c["noiseType"], c["hurst"], c["octaves"], c["xpos"], c["zpos"], c["noiseSize"]) == ("Cellular", 1.0, 13, 0, 0, 1757)
    out = {}
    out["simplex"] = oracle.fractal(oracle.SIMPLEX, RES, RES, s["hurst"], 1.0, 2.0, 0.0, s["octaves"], s["xpos"], s["zpos"],
                                    s["noiseSize"])
    out["cellular"] = oracle.fractal(oracle.CELLULAR, RES, RES, c["hurst"], 1.0, 2.0, 0.0, c["octaves"], c["xpos"], c["zpos"],
                                     c["noiseSize"])
    for k in ("simplex", "cellular"):
        out[k + "_gauss"] = oracle.kernel_filter(out[k], oracle.GAUSS5_S1, s["filterIterations"])
        # norm range of the demo assets; r is blind to it as long as nothing clips
        out[k + "_flow"] = oracle.flowmap(out[k + "_gauss"], s["flowIterations"], 0.0, 0.005)
    return out


def np_fractal(noise, hurst, octaves, xpos, zpos, noise_size, res=RES):
    """numpy restatement of FractalGenerator.NoiseValue (Noise/Fractal/Fractal.cs:109-131, amp 1, stepdown 2, no detune)
    around a basis from tests/np_noise.py -- only so that the basis can be MUTATED for the negative controls."""
    x = (np.arange(res, dtype=f32) + f32(xpos)) / f32(noise_size)
    z = (np.arange(res, dtype=f32) + f32(zpos)) / f32(noise_size)
    xi, zi = np.meshgrid(x, z)           # [z][x]
    G = f32(np.exp2(-np.float64(f32(hurst))))
    fr, a, t, norm = f32(1.0), f32(1.0), np.zeros((res, res), f32), f32(0.0)
    for _ in range(octaves):
        t = t + a * noise(fr * xi, fr * zi)
        norm = f32(norm + a)
        fr = f32(fr * f32(2.0))
        a = f32(a * G)
    return t / norm


def rectified_snoise(**mut):
    import np_noise as N
    return lambda x, z: (f32(1.0) + N.snoise2(x, z, **mut)) / f32(2.0)   # Fractal.cs:227-241


# ---- CPU: the oracle against the pictures ------------------------------------------------------------------------
def test_fixture_is_what_the_script_writes(shots):
    img, meta = shots
    assert sorted(img) == ["B2", "B5", "B6", "L0", "L1", "L3", "L4", "R2", "R5", "R6"]
    assert all(v.shape == (PIX, PIX) for v in img.values())
    assert all(m["resolution"] == RES and m["pixels"] == PIX for m in meta.values())


def stats(model, image, sigma=1.5):
    """(r, detail): Pearson r at the best translation, and r of the high-passed planes at that translation (sigma in
    fixture pixels; 1.5 keeps features below ~3 cells = fBm octaves 8 and up)."""
    r, tx, ty = fit(model, image)
    core = (slice(INSET, -INSET), slice(INSET, -INSET))
    w = _warp(ndimage.uniform_filter(np.asarray(model, np.float32), 2), tx, ty)
    return r, pearson(highpass(w, sigma)[core], highpass(image, sigma)[core])


def test_simplex_fbm_is_the_picture(planes, shots):
    img, _ = shots
    r, d = stats(display(planes["simplex"]), img["L3"])
    assert r >= 0.9995, r                                     # measured 0.99987
    assert d >= 0.9, d                                        # measured 0.926: the fine octaves are the picture's too
    # orientation and mapping were chosen on this image; other orientations are nowhere near
    wrong = [match(display(p), img["L3"]) for p in (planes["simplex"][:, ::-1], planes["simplex"][::-1], planes["simplex"].T)]
    assert max(wrong) < 0.6, wrong


def test_simplex_negative_controls_fall_below_the_bar(planes, shots, oracle):
    """The test has power: each single-constant mutation of the recalled snoise / of the inspector's parameters lands
    below what the true model reaches, r 0.99987 / detail 0.926 (the same translation fit is granted to each).
    Measured: hash multiplier 33 -> r -0.42; i1 inverted -> 0.833 / 0.50; C.y + 1 % -> 0.99915 / 0.84; zpos 0 -> 0.54;
    the README caption's 0.4 / 1700 -> 0.921; 8 octaves -> detail 0.68; Perlin -> 0.47.  Not resolvable at this pixel
    size: 12 vs 13 octaves (0.99984 / 0.923), hurst 0.43 (0.99983)."""
    img, meta = shots
    s = meta["3"]
    args = (s["hurst"], s["octaves"], s["xpos"], s["zpos"], s["noiseSize"])
    true_np = np_fractal(rectified_snoise(), *args)
    assert np.abs(true_np - planes["simplex"]).max() <= 2e-6          # the unmutated numpy form IS the oracle's plane
    assert match(display(true_np), img["L3"]) >= 0.9995

    def ofr(noise=oracle.SIMPLEX, hurst=s["hurst"], octaves=s["octaves"], zpos=s["zpos"], size=s["noiseSize"]):
        return oracle.fractal(noise, RES, RES, hurst, 1.0, 2.0, 0.0, octaves, s["xpos"], zpos, size)
    controls = {                                              # name: (plane, r bar, detail bar)
        "permute multiplier 33": (np_fractal(rectified_snoise(permute_mul=33.0), *args), 0.0, 0.3),
        "i1 select inverted": (np_fractal(rectified_snoise(invert_i1=True), *args), 0.9, 0.6),
        "C.y off by 1 %": (np_fractal(rectified_snoise(cy_scale=1.01), *args), 0.9995, 0.88),
        "zpos 0": (ofr(zpos=0), 0.6, 0.3),
        "README caption's hurst 0.4 / noiseSize 1700": (ofr(hurst=0.4, size=1700), 0.95, 0.5),
        "8 octaves": (ofr(octaves=8), 1.0, 0.75),
        "hurst 0.5": (ofr(hurst=0.5), 0.999, 1.0),
        "perlin basis": (ofr(noise=oracle.PERLIN), 0.6, 0.3),
    }
    for name, (plane, r_bar, d_bar) in controls.items():
        r, d = stats(display(plane), img["L3"])
        assert r < r_bar and d < d_bar, (name, r, d)


def test_cellular_fbm_is_the_picture(planes, shots, oracle):
    img, meta = shots
    r = match(display(planes["cellular"]), img["L0"])
    assert r >= 0.99, r
    # the README caption says Hurst 0.4; the inspector in the picture says 1 -- and the picture agrees with the inspector
    c = meta["0"]
    h04 = oracle.fractal(oracle.CELLULAR, RES, RES, 0.4, 1.0, 2.0, 0.0, c["octaves"], 0, 0, c["noiseSize"])
    assert match(display(h04), img["L0"]) < 0.95
    assert match(display(planes["simplex"]), img["L0"]) < 0.7   # a smooth picture: chance alone gives 0.5


def test_cellular_negative_controls_fall_below_the_bar(planes, shots):
    """The recalled `cellular` mutated in one constant: measured r 0.9947 for the spec, 0.689 with a hash
    multiplier of 33, 0.966 with jitter 0.8 (0.9885 with 0.9: the picture at hurst 1 is smooth, its power against small changes
    of this basis is limited) (rectified F1 * F2 per octave, Fractal.cs:263-278)."""
    import np_noise as N
    img, meta = shots
    c = meta["0"]
    args = (c["hurst"], c["octaves"], c["xpos"], c["zpos"], c["noiseSize"])

    def rectified(**mut):
        def basis(x, z):
            F1, F2 = N.cellular2(x, z, **mut)
            return ((f32(1.0) + F1) / f32(2.0)) * ((f32(1.0) + F2) / f32(2.0))
        return basis
    true_np = np_fractal(rectified(), *args)
    assert np.abs(true_np - planes["cellular"]).max() <= 2e-6
    assert match(display(true_np), img["L0"]) >= 0.99
    assert match(display(np_fractal(rectified(permute_mul=33.0), *args)), img["L0"]) < 0.8
    assert match(display(np_fractal(rectified(jitter=0.8), *args)), img["L0"]) < 0.99


def test_gauss5_x17_is_the_picture(planes, shots, oracle):
    img, _ = shots
    for basis, before, after in (("simplex", "L3", "L4"), ("cellular", "L0", "L1")):
        raw, flt = display(planes[basis]), display(planes[basis + "_gauss"])
        r_after, r_after_raw = match(flt, img[after]), match(raw, img[after])
        r_before, r_before_flt = match(raw, img[before]), match(flt, img[before])
        assert r_after >= (0.999 if basis == "simplex" else 0.99), (basis, r_after)
        if basis == "simplex":   # the cellular plane at hurst 1 has nothing for a radius-2 blur to remove (r differs in the 5th digit)
            assert r_after > r_after_raw + 0.003 and r_before > r_before_flt + 0.003, (r_after, r_after_raw, r_before, r_before_flt)
    # the detail the filter leaves (high-pass, sigma 3 fixture pixels): measured 0.70 / 0.78 / 0.89 / 0.943 / 0.86 for
    # 0 / 1 / 5 / 17 / 40 applications -- the picture was filtered about 12-25 times, as its inspector says
    d = {n: stats(display(oracle.kernel_filter(planes["simplex"], oracle.GAUSS5_S1, n) if n else planes["simplex"]),
                  img["L4"], 3.0)[1] for n in (0, 1, 5, 17, 40)}
    assert d[17] >= 0.93 and d[0] < d[1] < d[5] < d[17] - 0.03 and d[40] < d[17] - 0.05, d


def flow_display(v):
    return ndimage.gaussian_filter(np.asarray(v, np.float32)[:, ::-1], 1.0)


def test_flow_map_is_the_picture(planes, shots, oracle):
    img, _ = shots
    # the height underlay (R) is the Gauss-filtered plane in all three blue captures
    assert match(display(planes["simplex_gauss"]), img["R5"]) >= 0.998
    assert match(display(planes["simplex_gauss"]), img["R6"]) >= 0.998
    assert match(display(planes["cellular_gauss"]), img["R2"]) >= 0.98
    # the overlay (B) is the flow map, 5 iterations, of that plane
    r5 = match(flow_display(planes["simplex_flow"]), img["B5"])
    assert r5 >= 0.9, r5
    g = planes["simplex_gauss"]
    gz, gx = np.gradient(g)
    controls = {
        "flow of the unfiltered plane": oracle.flowmap(planes["simplex"], 5, 0.0, 0.005),
        "1 iteration": oracle.flowmap(g, 1, 0.0, 0.005),
        "12 iterations": oracle.flowmap(g, 12, 0.0, 0.005),
        "slope instead of flow": np.hypot(gx, gz).astype(f32),
        "flow of the transposed plane": oracle.flowmap(np.ascontiguousarray(g.T), 5, 0.0, 0.005),
    }
    got = {k: match(flow_display(v), img["B5"]) for k, v in controls.items()}
    bars = {"flow of the unfiltered plane": 0.5, "1 iteration": 0.75, "12 iterations": 0.85, "slope instead of flow": 0.5,
            "flow of the transposed plane": 0.3}
    for k in controls:
        assert got[k] < bars[k] and got[k] < r5 - 0.05, (k, got, r5)
    # cellular terrain: the first iteration is the picture (see the module docstring for why later ones are not)
    r2 = match(flow_display(oracle.flowmap(planes["cellular_gauss"], 1, 0.0, 0.005)), img["B2"])
    assert r2 >= 0.85, r2
    assert match(flow_display(planes["simplex_flow"]), img["B2"]) < 0.3


# ---- GPU: the HIP path, through the stages, against the same pictures ------------------------------------------------
def _hip_planes(nj, ctx, noise, hurst, octaves, xpos, zpos, noise_size, g_iter, f_iter):
    data = ctx.alloc(RES * RES)
    out = {}

    def run(stage):
        stage.ReceiveHandledInput(nj.PipelineWorkItem(nj.GeneratorData("shot", data, RES, xpos, zpos)), nj.JobHandle())
        stage.jobHandle.Complete()
        return data.ToArray((RES, RES)).copy()
    out["noise"] = run(nj.NoiseStage(ctx, noise, hurst, 1.0, octaves, 2.0, 0.0, noise_size))
    out["gauss"] = run(nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, g_iter))
    out["flow"] = run(nj.FlowMapStage(ctx, f_iter, 0.0, 0.005))
    data.Dispose()
    return out


@pytest.mark.gpu
def test_hip_path_is_the_pictures(nj, ctx, shots, planes):
    img, meta = shots
    s, c = meta["3"], meta["0"]
    hs = _hip_planes(nj, ctx, nj.FractalNoise.Simplex, s["hurst"], s["octaves"], s["xpos"], s["zpos"], s["noiseSize"],
                     s["filterIterations"], s["flowIterations"])
    hc = _hip_planes(nj, ctx, nj.FractalNoise.Cellular, c["hurst"], c["octaves"], c["xpos"], c["zpos"], c["noiseSize"],
                     c["filterIterations"], 1)
    assert match(display(hs["noise"]), img["L3"]) >= 0.999
    assert match(display(hs["gauss"]), img["L4"]) >= 0.999
    assert match(display(hs["noise"]), img["L4"]) < match(display(hs["gauss"]), img["L4"]) - 0.003
    assert match(flow_display(hs["flow"]), img["B5"]) >= 0.9
    assert match(display(hc["noise"]), img["L0"]) >= 0.99
    assert match(display(hc["gauss"]), img["L1"]) >= 0.99
    assert match(flow_display(hc["flow"]), img["B2"]) >= 0.85
    # and the planes the pictures were compared with are the oracle's, bit for bit (strict float mode)
    assert np.array_equal(hs["noise"], planes["simplex"]) and np.array_equal(hs["gauss"], planes["simplex_gauss"])
    assert np.array_equal(hs["flow"], planes["simplex_flow"]) and np.array_equal(hc["noise"], planes["cellular"])


@pytest.mark.gpu
def test_hip_path_in_fast_mode_is_the_pictures_too(nj, shots, planes):
    """NZ_FLOAT_FAST (the reference's own FloatMode.Fast: contracted tails, every stage within 1e-5 of strict) against the same
    pictures: the image pin does not depend on the strict build."""
    img, meta = shots
    s = meta["3"]
    with nj.Context(0) as c:
        c.float_mode = 1
        hs = _hip_planes(nj, c, nj.FractalNoise.Simplex, s["hurst"], s["octaves"], s["xpos"], s["zpos"], s["noiseSize"],
                         s["filterIterations"], s["flowIterations"])
    assert match(display(hs["noise"]), img["L3"]) >= 0.9995
    assert match(display(hs["gauss"]), img["L4"]) >= 0.999
    assert match(flow_display(hs["flow"]), img["B5"]) >= 0.9
    assert not np.array_equal(hs["noise"], planes["simplex"])           # (it IS the tolerance build)
    assert np.abs(hs["noise"] - planes["simplex"]).max() <= 1e-5 * np.abs(planes["simplex"]).max() + 1e-6
