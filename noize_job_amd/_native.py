"""ctypes binding of libnoize_hip.so (the C ABI declared in include/noize_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If the shared object is missing
or cannot be loaded, importing this module raises -- build it with `python -c "import
__graft_entry__ as g; g.build()"` or `make -C noize_job_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnoize_hip.so")

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
handle_t = C.c_uint64
handle_p = C.POINTER(C.c_uint64)
ctx_p = C.c_void_p
dev_ptr = C.c_void_p  # device addresses travel as plain integers


class Stripe(C.Structure):
    """nz_stripe (include/noize_hip.h)."""
    _fields_ = [("cols", C.c_int32), ("rows", C.c_int32), ("grow0", C.c_int32), ("grows", C.c_int32),
                ("own0", C.c_int32), ("own1", C.c_int32), ("pitch", C.c_int32)]


stripe_p = C.POINTER(Stripe)


class RWTile(C.Structure):
    """nz_rw_tile (include/noize_hip.h): the READ / WRITE slice pair of a tile (RWTileData, TileData.cs:49-93)."""
    _fields_ = [("read", C.c_void_p), ("write", C.c_void_p), ("resolution", C.c_int32), ("count", C.c_int32)]


rw_tile_p = C.POINTER(RWTile)


class ErosionParameters(C.Structure):
    """nz_erosion_params = ErosionParameters (Geologic/ParticleErosion/LiveErosionDataTypes.cs:78-100)."""
    _fields_ = ([(n, C.c_float) for n in ("INERTIA", "GRAVITY", "DRAG", "FRICTION", "EVAP", "EROSION", "DEPOSITION",
                                          "FLOW_HEIGHT_CONTRIBUTION", "SLOW_CULL_ANGLE", "SLOW_CULL_SPEED", "CAPACITY")] +
                [("MAXAGE", C.c_int32), ("TERMINAL_VELOCITY", C.c_float), ("SURFACE_EVAPORATION_RATE", C.c_float),
                 ("POOL_PLACEMENT_MULTIPLIER", C.c_float), ("TRACK_PLACEMENT_MULTIPLIER", C.c_float),
                 ("FLOW_LOSS_RATE", C.c_float), ("PILING_RADIUS", C.c_int32), ("MIN_PILE_INCREMENT", C.c_float),
                 ("PILE_THRESHOLD", C.c_float)])


class TileSetMeta(C.Structure):
    """nz_tile_set_meta = TileSetMeta (Pipeline/Tiles/TileTypes.cs:15-27)."""
    _fields_ = [("TILE_RES", C.c_int32 * 2), ("TILE_SIZE", C.c_int32 * 2), ("GENERATOR_RES", C.c_int32 * 2),
                ("PATCH_RES", C.c_float * 2), ("HEIGHT", C.c_int32), ("HEIGHT_F", C.c_float), ("MARGIN", C.c_int32)]


class Particle(C.Structure):
    """nz_particle: a queued BeyerParticle."""
    _fields_ = [("px", C.c_int32), ("pz", C.c_int32), ("water", C.c_float), ("pid", C.c_uint32)]


class TerrainParams(C.Structure):
    """nz_terrain_params (include/noize_hip.h): the stock stage list as a parameter block (nz_sharded_create)."""
    _fields_ = [("noiseType", C.c_int32), ("hurst", C.c_float), ("startingAmplitude", C.c_float), ("stepdown", C.c_float),
                ("detuneRate", C.c_float), ("octaves", C.c_int32), ("noiseSize", C.c_int32), ("filter", C.c_int32),
                ("filterIterations", C.c_int32), ("flowIterations", C.c_int32), ("normMin", C.c_float),
                ("normMax", C.c_float), ("erosionIterations", C.c_int32)]


class ShardedDesc(C.Structure):
    """nz_sharded_desc (include/noize_hip.h): one grid cut into row stripes over the ranks of a node."""
    _fields_ = [("grows", C.c_int32), ("cols", C.c_int32), ("stripes", C.c_int32), ("haloMode", C.c_int32),
                ("overlap", C.c_int32), ("xpos", C.c_int32), ("zpos", C.c_int32), ("externalSource", C.c_int32),
                ("asRank", C.c_int32), ("asWorld", C.c_int32)]


ep_p, tm_p, tp_p = C.POINTER(ErosionParameters), C.POINTER(TileSetMeta), C.POINTER(TerrainParams)
sd_p = C.POINTER(ShardedDesc)

NZ_OK, NZ_ERR_INVALID, NZ_ERR_UNSUPPORTED, NZ_ERR_HIP, NZ_ERR_NOMEM, NZ_ERR_NO_DEVICE, NZ_ERR_COMM, NZ_ERR_RETRY = 0, -1, -2, -3, -4, -5, -6, -7
NZ_COMM_ID_BYTES = 128
NZ_HALO_RECOMPUTE, NZ_HALO_EXCHANGE, NZ_HALO_EXCHANGE_ONCE = 0, 1, 2
NZ_FLOAT_STRICT, NZ_FLOAT_FAST, NZ_FLOAT_RELAXED = 0, 1, 2

_i, _f, _sz = C.c_int32, C.c_float, C.c_size_t
_tail = [handle_t, handle_p]  # (dep, out)

# name -> (restype, argtypes); every symbol include/noize_hip.h declares
SIGNATURES = {
    "nz_version": (_i, []),
    "nz_last_error": (C.c_char_p, []),
    "nz_device_count": (_i, [C.POINTER(_i)]),
    "nz_ctx_create": (_i, [_i, C.POINTER(ctx_p)]),
    "nz_ctx_create_on_stream": (_i, [_i, C.c_void_p, C.POINTER(ctx_p)]),
    "nz_ctx_destroy": (_i, [ctx_p]),
    "nz_ctx_synchronize": (_i, [ctx_p]),
    "nz_ctx_stream": (C.c_void_p, [ctx_p]),
    "nz_ctx_device": (_i, [ctx_p]),
    "nz_ctx_set_float_mode": (_i, [ctx_p, _i]),
    "nz_ctx_float_mode": (_i, [ctx_p]),
    "nz_tile_alloc": (_i, [ctx_p, _sz, C.POINTER(dev_ptr)]),
    "nz_tile_free": (_i, [ctx_p, dev_ptr]),
    "nz_tile_upload": (_i, [ctx_p, dev_ptr, C.c_void_p, _sz] + _tail),
    "nz_tile_download": (_i, [ctx_p, dev_ptr, C.c_void_p, _sz] + _tail),
    "nz_bytes_download": (_i, [ctx_p, dev_ptr, C.c_void_p, _sz] + _tail),
    "nz_flush_write_slice": (_i, [ctx_p, dev_ptr, dev_ptr, _sz] + _tail),
    "nz_handle_record": (_i, [ctx_p, handle_p]),
    "nz_handle_query": (_i, [ctx_p, handle_t, C.POINTER(_i)]),
    "nz_handle_wait": (_i, [ctx_p, handle_t]),
    "nz_handle_elapsed_ms": (_i, [ctx_p, handle_t, handle_t, C.POINTER(_f)]),
    "nz_handle_combine": (_i, [ctx_p, handle_p, _i, handle_p]),
    "nz_ctx_id": (_i, [ctx_p]),
    "nz_handle_context_id": (_i, [handle_t]),
    "nz_fractal": (_i, [ctx_p, _i, dev_ptr, _i, _f, _f, _f, _f, _i, _i, _i, _i] + _tail),
    "nz_fractal_stripe": (_i, [ctx_p, _i, dev_ptr, stripe_p, _f, _f, _f, _f, _i, _i, _i, _i] + _tail),
    "nz_kernel_filter": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i] + _tail),
    "nz_edge_1d_filter": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, _i] + _tail),
    "nz_edge_2d_filter": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i] + _tail),
    "nz_gauss_filter": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, _i] + _tail),
    "nz_smooth_filter": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i] + _tail),
    "nz_separable_series": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, f32p, f32p, _f] + _tail),
    "nz_erosion_kernel": (_i, [ctx_p, dev_ptr, _i] + _tail),
    "nz_kernel_filter_stage": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, _i] + _tail),
    "nz_gauss_blur_stage": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, _i, _i] + _tail),
    "nz_smooth_blur_stage": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, _i] + _tail),
    "nz_erosion_stage": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i] + _tail),
    "nz_debug_chain_delay": (_i, [_i, _i]),
    "nz_ctx_set_pile_safe": (_i, [ctx_p, _i]),
    "nz_ctx_pile_retries": (_i, [ctx_p]),
    "nz_debug_pile_poll_limit": (_i, [_i]),
    "nz_kernel_filter_halo_rows": (_i, [_i, _i]),
    "nz_kernel_filter_max_fused": (_i, [_i]),
    "nz_erosion_max_fused_iterations": (_i, []),
    "nz_kernel_filter_stripe": (_i, [ctx_p, dev_ptr, dev_ptr, stripe_p, _i, _i] + _tail),
    "nz_erosion_stripe": (_i, [ctx_p, dev_ptr, dev_ptr, stripe_p, _i] + _tail),
    "nz_fill_array": (_i, [ctx_p, dev_ptr, _i, _f] + _tail),
    "nz_flowmap_compute_flow": (_i, [ctx_p] + [dev_ptr] * 10 + [_i] + _tail),
    "nz_flowmap_update_water": (_i, [ctx_p] + [dev_ptr] * 6 + [_i] + _tail),
    "nz_flowmap_write_values": (_i, [ctx_p] + [dev_ptr] * 5 + [_i] + _tail),
    "nz_map_normalize_values": (_i, [ctx_p, dev_ptr, dev_ptr, f32p, _i] + _tail),
    "nz_get_map_range": (_i, [ctx_p, dev_ptr, C.c_size_t, dev_ptr, C.c_float, C.c_float] + _tail),
    "nz_map_normalize_values_dev": (_i, [ctx_p, dev_ptr, dev_ptr, dev_ptr, _i] + _tail),
    "nz_normalize_cells_dev": (_i, [ctx_p, dev_ptr, C.c_size_t, dev_ptr] + _tail),
    "nz_flowmap_stage_work_floats": (_sz, [_i]),
    "nz_flowmap_stage": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _f, _f, _i] + _tail),
    "nz_flow_fused_max_iterations": (_i, []),
    "nz_flow_fused_stripe": (_i, [ctx_p, dev_ptr, C.POINTER(dev_ptr), C.POINTER(dev_ptr), dev_ptr, stripe_p, _i, _i, _i,
                                  _f, _f] + _tail),
    "nz_constant_job": (_i, [ctx_p, _i, dev_ptr, dev_ptr, _f, _i] + _tail),
    "nz_reduction_job": (_i, [ctx_p, _i, dev_ptr, dev_ptr, dev_ptr, _i] + _tail),
    "nz_update_flow_from_track": (_i, [ctx_p, dev_ptr, dev_ptr, dev_ptr, _f, _f, _f, _i] + _tail),
    "nz_pool_automata": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i] + _tail),
    "nz_particle_queue_create": (_i, [ctx_p, _i, C.POINTER(C.c_void_p)]),
    "nz_particle_queue_destroy": (_i, [ctx_p, C.c_void_p]),
    "nz_particle_queue_count": (_i, [ctx_p, C.c_void_p, C.POINTER(_i)]),
    "nz_particle_queue_download": (_i, [ctx_p, C.c_void_p, C.c_void_p, _i, C.POINTER(_i)]),
    "nz_particle_queue_upload": (_i, [ctx_p, C.c_void_p, C.c_void_p, _i]),
    "nz_clear_particle_queue": (_i, [ctx_p, C.c_void_p] + _tail),
    "nz_erosive_events_create": (_i, [ctx_p, _i, C.POINTER(C.c_void_p)]),
    "nz_erosive_events_destroy": (_i, [ctx_p, C.c_void_p]),
    "nz_erosive_events_sediment": (C.c_void_p, [C.c_void_p]),
    "nz_erosive_events_count": (_i, [ctx_p, C.c_void_p, C.POINTER(_i)]),
    "nz_fill_beyer_queue": (_i, [ctx_p, C.c_void_p, ep_p, tm_p, _i, _i, _i, _i, _i] + _tail),
    "nz_queued_beyer_cycle": (_i, [ctx_p, dev_ptr, dev_ptr, dev_ptr, dev_ptr, C.c_void_p, C.c_void_p, ep_p, tm_p, _i, _i] + _tail),
    "nz_process_beyer_erosive_events": (_i, [ctx_p, dev_ptr, dev_ptr, dev_ptr, dev_ptr, C.c_void_p, ep_p, tm_p, _i] + _tail),
    "nz_erode_height_maps": (_i, [ctx_p, dev_ptr, C.c_void_p, ep_p, tm_p, _i] + _tail),
    "nz_erode_height_maps_and_flow": (_i, [ctx_p, dev_ptr, C.c_void_p, dev_ptr, dev_ptr, dev_ptr, ep_p, tm_p, _i] + _tail),
    "nz_pool_automata_job": (_i, [ctx_p, dev_ptr, dev_ptr, C.c_void_p, ep_p, tm_p, _i, _i, _i] + _tail),
    "nz_curviture_map": (_i, [ctx_p, dev_ptr, dev_ptr, tm_p, _i, _i, _i] + _tail),
    "nz_set_rgba32": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, _i, _f] + _tail),
    "nz_crop_job": (_i, [ctx_p, dev_ptr, _i, dev_ptr, _i] + _tail),
    "nz_curve_job": (_i, [ctx_p, dev_ptr, dev_ptr, dev_ptr, _i, _i] + _tail),
    "nz_thermal_erosion": (_i, [ctx_p, dev_ptr, _f, _f, _f, _i, _i] + _tail),
    "nz_fractal_batch": (_i, [ctx_p, _i, dev_ptr, _i, _i, dev_ptr, _f, _f, _f, _f, _i, _i] + _tail),
    "nz_kernel_filter_stage_batch": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, _i, _i] + _tail),
    "nz_gauss_blur_stage_batch": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, _i, _i, _i] + _tail),
    "nz_smooth_blur_stage_batch": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, _i, _i] + _tail),
    "nz_kernel_filter_stage_rw": (_i, [ctx_p, rw_tile_p, _i, _i] + _tail),
    "nz_gauss_blur_stage_rw": (_i, [ctx_p, rw_tile_p, _i, _i, _i] + _tail),
    "nz_smooth_blur_stage_rw": (_i, [ctx_p, rw_tile_p, _i, _i] + _tail),
    "nz_erosion_stage_rw": (_i, [ctx_p, rw_tile_p, _i] + _tail),
    "nz_flowmap_stage_rw_work_floats": (_sz, [_i, _i]),
    "nz_flowmap_stage_rw": (_i, [ctx_p, rw_tile_p, dev_ptr, _i, _f, _f] + _tail),
    "nz_erosion_stage_batch": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _i, _i] + _tail),
    "nz_flowmap_stage_batch": (_i, [ctx_p, dev_ptr, dev_ptr, _i, _f, _f, _i, _i] + _tail),
    "nz_mesh_vertex_count": (_sz, [_i]),
    "nz_mesh_index_count": (_sz, [_i]),
    "nz_heightmap_mesh": (_i, [ctx_p, _i, dev_ptr, dev_ptr, _i, _i, _i, _f, _f, dev_ptr] + _tail),
    "nz_heightmap_mesh16": (_i, [ctx_p, _i, dev_ptr, dev_ptr, _i, _i, _i, _f, _f, dev_ptr] + _tail),
    "nz_square_grid_mesh": (_i, [ctx_p, dev_ptr, dev_ptr, _i] + _tail),
    "nz_heightmap_mesh_batch": (_i, [ctx_p, _i, dev_ptr, dev_ptr, _i, _i, _i, _f, _f, dev_ptr, _i] + _tail),
    "nz_debug_chain_poll_limit": (_i, [_i]),
    # one grid over the GPUs of a node (nz_comm.cpp)
    "nz_comm_unique_id": (_i, [C.c_void_p]),
    "nz_comm_init": (_i, [ctx_p, C.c_void_p, _i, _i, C.POINTER(C.c_void_p)]),
    "nz_comm_destroy": (_i, [C.c_void_p]),
    "nz_comm_rank": (_i, [C.c_void_p]),
    "nz_comm_world": (_i, [C.c_void_p]),
    "nz_comm_rccl_version": (_i, [C.POINTER(_i)]),
    "nz_halo_exchange_begin": (_i, [ctx_p, C.c_void_p, C.POINTER(dev_ptr), _i, stripe_p, _i, _i, handle_t]),
    "nz_halo_exchange_finish": (_i, [ctx_p, C.c_void_p, handle_p]),
    "nz_halo_exchange": (_i, [ctx_p, C.c_void_p, C.POINTER(dev_ptr), _i, stripe_p, _i, _i] + _tail),
    "nz_comm_allgather_range": (_i, [ctx_p, C.c_void_p, dev_ptr, _sz, dev_ptr, _f, _f] + _tail),
    "nz_sharded_create": (_i, [ctx_p, C.c_void_p, sd_p, tp_p, C.POINTER(C.c_void_p)]),
    "nz_sharded_destroy": (_i, [C.c_void_p]),
    "nz_sharded_local_stripes": (_i, [C.c_void_p]),
    "nz_sharded_stripe": (_i, [C.c_void_p, _i, stripe_p, C.POINTER(dev_ptr), C.POINTER(dev_ptr)]),
    "nz_sharded_plan": (_i, [C.c_void_p, C.POINTER(_i), _i, C.POINTER(_i)]),
    "nz_sharded_transfers": (_i, [C.c_void_p, C.POINTER(_i), _i, C.POINTER(_i)]),
    "nz_sharded_pipeline": (_i, [ctx_p, C.c_void_p, handle_p] + _tail),
    "nz_sharded_traffic": (_i, [C.c_void_p, C.POINTER(_i), C.POINTER(_sz)]),
    "nz_sharded_set_timing": (_i, [C.c_void_p, _i]),
    "nz_sharded_exchange_ms": (_i, [C.c_void_p, C.POINTER(_f)]),
    "nz_sharded_map_range": (_i, [ctx_p, C.c_void_p, dev_ptr, _f, _f] + _tail),
    "nz_sharded_normalize": (_i, [ctx_p, C.c_void_p, dev_ptr] + _tail),
}


def _share_hip_runtime_with_torch():
    """One HIP runtime per process: KFD admits a single compute context per process, and the PyTorch
    wheel bundles its own libamdhip64 (same SONAME as /opt/rocm's, but requested by torch under the
    unversioned name, so the loader would map a second copy if ours were resident first and that
    copy finds no device).  When torch is installed, map its copy first; libnoize_hip.so then binds to
    it by SONAME and torch, RCCL and this library share streams and allocations.  Without torch the
    system runtime in /opt/rocm is used."""
    import importlib.util
    import sys
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    wheel_hip = "torch" in sys.modules  # torch has mapped its own runtime already
    if not wheel_hip:
        cand = os.path.join(libdir, "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                wheel_hip = True
            except OSError:
                pass
    # RCCL is opened lazily by the library (nz_comm.cpp, first communicator call).  Point it at the wheel's copy only when the
    # wheel's HIP runtime is the one this process runs on -- an RCCL built against another runtime than the one
    # libnoize_hip.so is bound to must not be mixed in; otherwise the system's librccl.so.1 is found by name
    rccl = os.path.join(libdir, "librccl.so")
    if wheel_hip and os.path.exists(rccl):
        os.environ.setdefault("NZ_RCCL_LIB", rccl)


def _load():
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "noize_job_amd: %s is missing -- the HIP extension is not built (run __graft_entry__.build()); "
            "there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class NoizeError(RuntimeError):
    """The C# host maps negative statuses to exceptions (error convention)."""

    def __init__(self, status, where):
        self.status = status
        msg = lib.nz_last_error()
        super().__init__("%s failed with status %d: %s" % (where, status, msg.decode() if msg else ""))


def check(status, where):
    if status != NZ_OK:
        raise NoizeError(status, where)
