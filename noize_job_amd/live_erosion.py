"""Host-side mirror of the reference's live particle erosion driver (BASELINE config 4).

Geologic/ParticleErosion/Component/LiveErosion.cs (the MonoBehaviour that owns the planes and chains the jobs,
TriggerQueuedBeyerMT :378-436) and ScriptableObject/ErosionSettings.cs, minus Unity: the planes live in HBM, every job
is one C-ABI call on the context's stream, and the random seed of each cycle is an argument (the reference draws it
from UnityEngine.Random, MultiThreadErosionJob.cs:50).  Same seeds, same planes -> the same result on every run, bit
for bit (see noize_job_amd/csrc/nz_live.hip for the three choices that are fixed).
"""
import ctypes as C
import enum

import numpy as np

from . import _native as N
from .runtime import JobHandle


class ErosionMode(enum.IntEnum):  # LiveErosionDataTypes.cs:29-34
    ALL_EROSION = 0
    ONLY_THERMAL_EROSION = 1
    THERMAL_FLOW_WATER = 2
    ONLY_FLOW_WATER = 3


class ColorChannelByte(enum.IntEnum):  # LiveErosionDataTypes.cs:1235-1241
    R = 0
    G = 1
    B = 2
    A = 3


class ErosionSettings:  # ScriptableObject/ErosionSettings.cs:5-124 (defaults = Reset())
    def __init__(self, **kw):
        self.CYCLES = 3
        self.PARTICLES_PER_CYCLE = 1000
        self.BEHAVIOR = ErosionMode.ALL_EROSION
        self.INERTIA = 0.5
        self.GRAVITY = 1.0
        self.DRAG = 0.001
        self.FRICTION = 0.01
        self.EVAP = 0.01
        self.EROSION = 1.0
        self.DEPOSITION = 0.1
        self.FLOW_HEIGHT_CONTRIBUTION = 25.0
        self.SLOW_CULL_ANGLE = 3.0
        self.SLOW_CULL_SPEED = 0.11
        self.CAPACITY = 3.0
        self.MAXAGE = 100
        self.WATER_STEPS = 10
        self.SURFACE_EVAPORATION_RATE = 0.1
        self.POOL_PLACEMENT_MULTIPLIER = 0.5
        self.TRACK_PLACEMENT_MULTIPLIER = 80.0
        self.FLOW_LOSS_RATE = 0.05
        self.PILING_RADIUS = 15
        self.MIN_PILE_INCREMENT = 1.0
        self.PILE_THRESHOLD = 2.0
        self.ENABLE_THERMAL = True
        self.TALUS = 55.0
        self.THERMAL_STEP = .6
        self.THERMAL_CYCLES = 1
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("ErosionSettings has no field %s" % k)
            setattr(self, k, v)

    def AsParameters(self):  # :96-123
        ep = N.ErosionParameters()
        for name in ("INERTIA", "GRAVITY", "FRICTION", "DRAG", "EVAP", "EROSION", "DEPOSITION", "FLOW_HEIGHT_CONTRIBUTION",
                     "SLOW_CULL_ANGLE", "SLOW_CULL_SPEED", "MAXAGE", "SURFACE_EVAPORATION_RATE", "TRACK_PLACEMENT_MULTIPLIER",
                     "FLOW_LOSS_RATE", "PILING_RADIUS", "MIN_PILE_INCREMENT", "PILE_THRESHOLD"):
            setattr(ep, name, getattr(self, name))
        ep.CAPACITY = self.CAPACITY if self.BEHAVIOR == ErosionMode.ALL_EROSION else 0
        ep.TERMINAL_VELOCITY = float(np.float32(1.0) / np.float32(self.DRAG))
        ep.POOL_PLACEMENT_MULTIPLIER = 0.0 if self.BEHAVIOR == ErosionMode.ONLY_THERMAL_EROSION else self.POOL_PLACEMENT_MULTIPLIER
        return ep


def tile_set_meta(generator_res, height=1000, tile_size=1000, tile_res=None, margin=0, patch_res=None):
    """TileSetMeta (Pipeline/Tiles/TileTypes.cs:15-27); PATCH_RES defaults to TILE_SIZE / TILE_RES."""
    tile_res = generator_res - 2 * margin if tile_res is None else tile_res
    tm = N.TileSetMeta()
    tm.TILE_RES[0] = tm.TILE_RES[1] = tile_res
    tm.TILE_SIZE[0] = tm.TILE_SIZE[1] = tile_size
    tm.GENERATOR_RES[0] = tm.GENERATOR_RES[1] = generator_res
    pr = float(np.float32(tile_size) / np.float32(tile_res)) if patch_res is None else patch_res
    tm.PATCH_RES[0] = tm.PATCH_RES[1] = pr
    tm.HEIGHT = height
    tm.HEIGHT_F = float(height)
    tm.MARGIN = margin
    return tm


class ParticleQueue:
    """NativeQueue<BeyerParticle> in device memory (nz_particle_queue)."""
    DTYPE = np.dtype([("px", np.int32), ("pz", np.int32), ("water", np.float32), ("pid", np.uint32)])

    def __init__(self, ctx, capacity):
        self.ctx, self.capacity = ctx, int(capacity)
        h = C.c_void_p()
        N.check(N.lib.nz_particle_queue_create(ctx._h, self.capacity, C.byref(h)), "nz_particle_queue_create")
        self._h = h

    @property
    def Count(self):
        n = C.c_int32(0)
        N.check(N.lib.nz_particle_queue_count(self.ctx._h, self._h, C.byref(n)), "nz_particle_queue_count")
        return n.value

    def ToArray(self):
        out = np.zeros(self.capacity, self.DTYPE)
        n = C.c_int32(0)
        N.check(N.lib.nz_particle_queue_download(self.ctx._h, self._h, out.ctypes.data, self.capacity, C.byref(n)),
                "nz_particle_queue_download")
        return out[:n.value].copy()

    def CopyFrom(self, particles):
        p = np.ascontiguousarray(particles, self.DTYPE)
        N.check(N.lib.nz_particle_queue_upload(self.ctx._h, self._h, p.ctypes.data, len(p)), "nz_particle_queue_upload")

    def Clear(self, dep=None, handle=True):
        return self.ctx.call("nz_clear_particle_queue", self._h, dep=dep, handle=handle)

    def Dispose(self):
        if self._h:
            N.check(N.lib.nz_particle_queue_destroy(self.ctx._h, self._h), "nz_particle_queue_destroy")
            self._h = None


class ErosiveEvents:
    """The events multi-hash-map and the erosions queue of LiveErosion (:225-229) as per-cell device planes."""

    def __init__(self, ctx, resolution):
        self.ctx, self.resolution = ctx, resolution
        h = C.c_void_p()
        N.check(N.lib.nz_erosive_events_create(ctx._h, resolution, C.byref(h)), "nz_erosive_events_create")
        self._h = h

    def sediment(self):
        """The per-cell ErosiveEvent.deltaSediment of the last ProcessBeyerErosiveEventsJob, [x, z]."""
        ptr = N.lib.nz_erosive_events_sediment(self._h)
        return self.ctx.wrap(ptr, self.resolution * self.resolution).ToArray((self.resolution, self.resolution))

    @property
    def Count(self):
        n = C.c_int32(0)
        N.check(N.lib.nz_erosive_events_count(self.ctx._h, self._h, C.byref(n)), "nz_erosive_events_count")
        return n.value

    def Dispose(self):
        if self._h:
            N.check(N.lib.nz_erosive_events_destroy(self.ctx._h, self._h), "nz_erosive_events_destroy")
            self._h = None


class LiveErosion:
    """Component/LiveErosion.cs without the MonoBehaviour: owns heightMap / poolMap / streamMap / particleTrack (planes
    indexed x * res + z), the particle queue, the events, and schedules one Update's worth of jobs."""

    def __init__(self, ctx, heightMap, tileMeta, erosionSettings=None, performErosion=True, queueCapacity=None):
        self.ctx = ctx
        self.tileMeta = tileMeta
        self.res = tileMeta.GENERATOR_RES[0]
        self.erosionSettings = erosionSettings if erosionSettings is not None else ErosionSettings()
        self.performErosion = performErosion
        n = self.res * self.res
        self.heightMap = heightMap  # DeviceTile
        assert heightMap.Length == n
        self.poolMap, self.streamMap, self.particleTrack = (ctx.from_host(np.zeros(n, np.float32)) for _ in range(3))
        self.QUEUE_SIZE = 1 if not performErosion else self.erosionSettings.PARTICLES_PER_CYCLE  # :215-219
        # drained pools add to the queue on top of QUEUE_SIZE
        self.particleQueue = ParticleQueue(ctx, queueCapacity or max(4 * self.QUEUE_SIZE, self.QUEUE_SIZE + n // 8, 1024))
        self.events = ErosiveEvents(ctx, self.res)
        self.particleGenerationID = 0
        self.EVENT_LIMIT = 1500
        self.jobHandle = JobHandle()
        self.waterControl = self.textureControl = None
        # ErodeHeightMaps and UpdateFlowFromTrackJob are siblings in the reference's job graph (CombineDependencies, :408-412):
        # one call, the pile solver's launch carries the flow update's workgroups (nz_erode_height_maps_and_flow).  False: the
        # two entries one after the other on the context's stream.  (Rounds 3 and 4 could run the second on a stream of its own:
        # at 8192^2 the two cross-stream dependencies cost more than the overlap saved, 2.71 against 2.54 ms per driver cycle.)
        self.fuseSiblings = True
        # The jobs of a cycle are links of ONE chain on this context's stream: only the handles somebody waits for are
        # asked of the library (the cycle chain's last) -- a handle is an event record, ~3 us of the stream.  False: one per job, as the reference
        # schedules them.
        self.fewHandles = True

    @property
    def safe(self):
        """nz_ctx_set_pile_safe: ErodeHeightMaps keeps a copy of the height plane, waits for the pile solver's one-launch form and
        runs itself again colour by colour should a block of it ever give up (include/noize_hip.h).  A property of the CONTEXT."""
        return getattr(self.ctx, "_pile_safe", False)

    @safe.setter
    def safe(self, on):
        N.check(N.lib.nz_ctx_set_pile_safe(self.ctx._h, int(bool(on))), "nz_ctx_set_pile_safe")
        self.ctx._pile_safe = bool(on)

    @property
    def pileRetries(self):
        return N.lib.nz_ctx_pile_retries(self.ctx._h)

    def _call(self, name, *args, dep=None, handle=True):
        return self.ctx.call(name, *args, dep=dep, handle=handle or not self.fewHandles)

    def TriggerQueuedBeyerMT(self, seeds):
        """:378-436.  `seeds`: one int per cycle (stands for UnityEngine.Random.Range in FillBeyerQueueJob)."""
        es, tm, res = self.erosionSettings, self.tileMeta, self.res
        ep = es.AsParameters()
        epp, tmp_ = C.byref(ep), C.byref(tm)
        handle = JobHandle()
        if self.performErosion:
            assert len(seeds) >= es.CYCLES, "one seed per cycle"
            for i in range(es.CYCLES):
                # the only handle of a cycle anybody outside this stream looks at is its last one -- and that only when
                # nothing follows on the stream
                last = i + 1 == es.CYCLES and self.waterControl is None
                if es.ENABLE_THERMAL and es.BEHAVIOR != ErosionMode.ONLY_FLOW_WATER:
                    handle = self._call("nz_thermal_erosion", self.heightMap.ptr, float(es.TALUS), float(es.THERMAL_STEP),
                                        float(tm.TILE_SIZE[0] // tm.HEIGHT),  # `TILE_SIZE.x / HEIGHT`: both int in C# (:386)
                                        es.THERMAL_CYCLES, res,
                                        dep=handle, handle=False)
                if es.BEHAVIOR != ErosionMode.ONLY_FLOW_WATER:
                    handle = self._call("nz_fill_beyer_queue", self.particleQueue._h, epp, tmp_, self.particleGenerationID % 4,
                                        res, self.QUEUE_SIZE, int(seeds[i]), min(10, self.QUEUE_SIZE), dep=handle, handle=False)
                # ClearQueueJob<ErosiveEvent> / ClearMultiDict: the event planes clear themselves when they are processed
                # CopyBeyerQueueJob: the queue's device array is the list
                handle = self._call("nz_queued_beyer_cycle", self.heightMap.ptr, self.poolMap.ptr, self.streamMap.ptr,
                                    self.particleTrack.ptr, self.particleQueue._h, self.events._h, epp, tmp_, self.EVENT_LIMIT,
                                    res, dep=handle, handle=False)
                handle = self._call("nz_process_beyer_erosive_events", self.heightMap.ptr, self.poolMap.ptr, self.streamMap.ptr,
                                    self.particleTrack.ptr, self.events._h, epp, tmp_, res, dep=handle, handle=False)
                # handle = CombineDependencies(ClearQueueJob, ErodeHeightMaps, UpdateFlowFromTrackJob), all three behind the
                # event reduction (:408-412)
                handle = self.particleQueue.Clear(dep=handle, handle=not self.fewHandles)
                if self.fuseSiblings:
                    handle = self._call("nz_erode_height_maps_and_flow", self.heightMap.ptr, self.events._h, self.poolMap.ptr,
                                        self.streamMap.ptr, self.particleTrack.ptr, epp, tmp_, res, dep=handle, handle=False)
                else:
                    handle = self._call("nz_erode_height_maps", self.heightMap.ptr, self.events._h, epp, tmp_, res, dep=handle,
                                        handle=False)
                    handle = self._call("nz_update_flow_from_track", self.poolMap.ptr, self.streamMap.ptr, self.particleTrack.ptr,
                                        ep.FLOW_LOSS_RATE, ep.SURFACE_EVAPORATION_RATE, float(tm.HEIGHT), res, dep=handle,
                                        handle=False)
                handle = self._call("nz_pool_automata_job", self.poolMap.ptr, self.heightMap.ptr, self.particleQueue._h, epp,
                                    tmp_, es.WATER_STEPS, res, int(self.performErosion), dep=handle, handle=last)
        if self.waterControl is not None:  # the RGBA32 control textures (:418-430)
            mres = self.tileMeta.TILE_RES[0]
            for src, tex, ch, scale in ((self.poolMap, self.waterControl, ColorChannelByte.R, 1000.0),
                                        (self.poolMap, self.waterControl, ColorChannelByte.G, 1000.0),
                                        (self.streamMap, self.waterControl, ColorChannelByte.B, 2.0),
                                        (self.streamMap, self.textureControl, ColorChannelByte.G, 3.0)):
                handle = self._call("nz_set_rgba32", src.ptr, tex.ptr, int(ch), res, mres, scale, dep=handle, handle=False)
            handle = self._call("nz_curviture_map", self.textureControl.ptr, self.heightMap.ptr, tmp_, int(ColorChannelByte.G),
                                res, mres, dep=handle, handle=False)
            handle = self._call("nz_set_rgba32", self.streamMap.ptr, self.textureControl.ptr, int(ColorChannelByte.A), res, mres,
                                1.0, dep=handle)
        self.jobHandle = handle
        self.particleGenerationID += 1  # Update() :341, once per completed job
        return handle

    def EnableControlTextures(self):
        mres = self.tileMeta.TILE_RES[0]
        self.waterControl = self.ctx.from_host(np.zeros(mres * mres * 4, np.uint8))
        self.textureControl = self.ctx.from_host(np.zeros(mres * mres * 4, np.uint8))

    def CompleteJob(self):
        """jobHandle.Complete() of Update() (:330-343), plus what a NativeQueue cannot do silently: a queue that a job found
        too small (drained pools or a top-up beyond its capacity drop particles, which changes the erosion) raises here."""
        self.jobHandle.Complete()
        self.particleQueue.Count  # NZ_ERR_NOMEM -> exception if a job overflowed the queue

    def OnDestroy(self):
        self.CompleteJob()
        for t in (self.poolMap, self.streamMap, self.particleTrack, self.waterControl, self.textureControl):
            if t is not None:
                t.Dispose()
        self.particleQueue.Dispose()
        self.events.Dispose()
