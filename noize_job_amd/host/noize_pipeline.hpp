// noize_pipeline.hpp -- C++ host side above the C ABI: the reference's Pipeline / PipelineStage operator
// API for the hot path, for hosts written in a compiled language (the reference's own host language,
// C#, has no toolchain in this image; its P/Invoke form is in INTEGRATION.md).
//
// Same class names, field names, argument meaning and error behaviour as
//   Pipeline/Stage/PipelineStage.cs:10-62, Pipeline/Stage/StageIO.cs:8-11,
//   Pipeline/Stage/StageIOTypes/{GeneratorData,MeshStageData}.cs, Pipeline/Stage/PipelineDefinition.cs:18-25,
//   Pipeline/Executable/Pipeline.cs:19-287, Noise/NoiseStage.cs:13-61, Filter/KernelFilterStage.cs:13-51,
//   Filter/Kernel/Blur/Stage{Gaussian,Smooth}Blur.cs, Geologic/Stage/FlowMapStage.cs:16-220,
//   Mesh/Stage/MeshTileStage.cs:28-61.
// Header-only; link with -lnoize_hip.
#pragma once

#include <algorithm>
#include <array>
#include <cstdio>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <vector>

#include "../../include/noize_hip.h"

namespace noize {

struct NoizeError : std::runtime_error {
    int32_t status;
    NoizeError(int32_t s, const std::string &where)
        : std::runtime_error(where + " failed with status " + std::to_string(s) + ": " + nz_last_error()), status(s) {}
};

inline void check(int32_t status, const char *where) {
    if (status != NZ_OK) throw NoizeError(status, where);
}

// Unity.Jobs.JobHandle
struct JobHandle {
    nz_ctx *ctx = nullptr;
    nz_handle id = 0;
    bool IsCompleted() const {
        if (!ctx || id == 0) return true;
        int32_t done = 0;
        check(nz_handle_query(ctx, id, &done), "nz_handle_query");
        return done != 0;
    }
    void Complete() const {
        if (ctx && id) check(nz_handle_wait(ctx, id), "nz_handle_wait");
    }
};

// NativeArray<float> / NativeSlice<float> in HBM
struct DeviceTile {
    nz_ctx *ctx = nullptr;
    float *ptr = nullptr;
    size_t Length = 0;
    bool owned = false;
    DeviceTile() = default;
    DeviceTile(nz_ctx *c, size_t n) : ctx(c), Length(n), owned(true) { check(nz_tile_alloc(c, n, &ptr), "nz_tile_alloc"); }
    DeviceTile(const DeviceTile &) = delete;
    DeviceTile &operator=(const DeviceTile &) = delete;
    ~DeviceTile() { Dispose(); }
    bool IsCreated() const { return ptr != nullptr; }
    void Dispose() {
        if (owned && ptr) nz_tile_free(ctx, ptr);
        ptr = nullptr;
    }
    void CopyFrom(const float *host) {
        check(nz_tile_upload(ctx, ptr, host, Length, 0, nullptr), "nz_tile_upload");
        check(nz_ctx_synchronize(ctx), "nz_ctx_synchronize");
    }
    void CopyTo(float *host) const {
        check(nz_tile_download(ctx, ptr, host, Length, 0, nullptr), "nz_tile_download");
        check(nz_ctx_synchronize(ctx), "nz_ctx_synchronize");
    }
};

// ---- StageIO payloads ------------------------------------------------------------------------
struct StageIO {
    std::string uuid;
    DeviceTile *data = nullptr;  // NativeSlice<float>
    virtual ~StageIO() = default;
};

struct GeneratorData : StageIO {
    int resolution = 512, xpos = 0, zpos = 0;
    // optional WRITE slice of the tile's RWTileData pair (Pipeline/Tiles/TileData.cs:49-93; new-framework): with it
    // the stencil stages run their nz_*_rw forms and TileHelpers.SWAP_RWTILE swaps `data` and `write` instead of
    // copying, so after a stage `data` is whichever of the two planes holds the result
    DeviceTile *write = nullptr;
};

struct GeneratorDataBatch;
// the payload's READ / WRITE pair as nz_rw_tile, and back (adopts the pair as the call left it)
inline nz_rw_tile rw_pair(GeneratorData *d, int count) { return nz_rw_tile{d->data->ptr, d->write->ptr, d->resolution, count}; }
inline void rw_adopt(GeneratorData *d, const nz_rw_tile &t) {
    if (t.read != d->data->ptr) std::swap(d->data, d->write);
}

// New-framework payload: `count` independent tiles of resolution^2 cells stored back to back in `data`,
// world positions {xpos, zpos} per tile in the device int32 array `positions` (see nz_*_batch).
struct GeneratorDataBatch : GeneratorData {
    int count = 1;
    const int32_t *positions = nullptr;
};

inline int tile_count(GeneratorData *d) {
    auto *b = dynamic_cast<GeneratorDataBatch *>(d);
    return b ? b->count : 1;
}

struct MeshBuffers {  // stands in for UnityEngine.Mesh + Mesh.MeshData (PositionStream32 layout)
    std::unique_ptr<DeviceTile> vertices, indices;
    size_t vertexCount = 0, indexCount = 0;
};

struct MeshStageData : StageIO {
    int resolution = 512, inputResolution = 512, marginPix = 5;
    float tileSize = 512.f, tileHeight = 512.f;
    int xpos = 0, zpos = 0;
    MeshBuffers *mesh = nullptr;
    int count = 1;  // new-framework: `count` height planes stored back to back -> `count` meshes (nz_heightmap_mesh_batch)
};

class PipelineStateManager;

struct PipelineWorkItem {
    StageIO *data = nullptr;
    std::function<void(StageIO *)> completeAction;
    std::function<void(StageIO *, JobHandle)> scheduledAction;
    JobHandle dependency;
    PipelineStateManager *stageManager = nullptr;  // Pipeline/Stage/PipelineDefinition.cs:18-25
};

struct DownsampleData : StageIO {  // StageIOTypes/DownsampleData.cs:9-17
    int resolution = 512, inputResolution = 512;
    DeviceTile *inputData = nullptr;
};

// ---- PipelineStateManager, NativeArray<float> part (Pipeline/PipelineState/PipelineStateManager.cs:13-189,
//      PipelineState.cs:230-349) with the job-fence lock of PipelineStateLock.cs:12-27: named device buffers shared
//      between pipelines, locked while a scheduled write to them has not completed ------------------------------
class PipelineStateManager {
  public:
    explicit PipelineStateManager(nz_ctx *c) : ctx(c) {}
    DeviceTile *GetBuffer(const std::string &name, long size = -1) {
        auto it = buffers.find(name);
        if (it == buffers.end()) {
            if (size < 0) throw std::invalid_argument("No allocated buffer named " + name);
            it = buffers.emplace(name, std::unique_ptr<DeviceTile>(new DeviceTile(ctx, (size_t)size))).first;
        }
        return it->second.get();
    }
    bool BufferExists(const std::string &name) const { return buffers.count(name) != 0; }
    bool ReleaseBuffer(const std::string &name) { return buffers.erase(name) != 0; }
    bool IsLocked(const std::string &key) const {
        auto it = locks.find(key);
        return it != locks.end() && !it->second.IsCompleted();  // HandleLock.isLocked
    }
    bool TrySetLock(const std::string &key, JobHandle handle, JobHandle /*spyHandle*/) {
        if (IsLocked(key)) return false;  // a completed handle is an open lock
        locks[key] = handle;
        return true;
    }
    void OnDestroy() { buffers.clear(); }

  private:
    nz_ctx *ctx;
    std::map<std::string, std::unique_ptr<DeviceTile>> buffers;
    std::map<std::string, JobHandle> locks;
};

// ---- PipelineStage ---------------------------------------------------------------------------
class BasePipeline;
class PipelineStage {
    friend class BasePipeline;  // the one-call form of the stock stage list sets the stages' handles itself

  public:
    explicit PipelineStage(nz_ctx *c) : ctx(c) {}
    virtual ~PipelineStage() = default;
    std::vector<std::function<void(PipelineWorkItem &, JobHandle)>> OnStageScheduledAction;

    virtual void ResizeNativeContainers(size_t) {}
    virtual bool IsSchedulable(const PipelineWorkItem &) { return true; }
    virtual void Schedule(PipelineWorkItem &, JobHandle) {}
    virtual void TransformData(PipelineWorkItem &) {}
    virtual void OnStageComplete() {}
    virtual void OnDestroy() {}
    nz_ctx *Context() const { return ctx; }

    template <class T>
    T *CheckRequirements(PipelineWorkItem &requirements) {
        T *d = dynamic_cast<T *>(requirements.data);
        if (!d) throw std::runtime_error("Unhandled stageio");  // PipelineStage.cs:37
        if (d->data->Length != dataLength) {
            dataLength = d->data->Length;
            ResizeNativeContainers(dataLength);
        }
        return d;
    }
    void ReceiveHandledInput(PipelineWorkItem &requirements, JobHandle dependency) {
        Schedule(requirements, dependency);
        TransformData(requirements);
        for (auto &a : OnStageScheduledAction) a(requirements, jobHandle);
    }

  protected:
    nz_ctx *ctx;
    JobHandle jobHandle;
    size_t dataLength = 0;
    JobHandle done(nz_handle h) { return JobHandle{ctx, h}; }
};

enum class FractalNoise { Sin, Perlin, PeriodicPerlin, Simplex, RotatedSimplex, Cellular, DomainRotatedPerlin, DomainRotatedSimplex };

class NoiseStage : public PipelineStage {
  public:
    using PipelineStage::PipelineStage;
    FractalNoise noiseType = FractalNoise::Sin;
    float hurst = 0.f, startingAmplitude = 1.f, stepdown = 2.f, detuneRate = 0.f;
    int octaves = 1, noiseSize = 1000;
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = CheckRequirements<GeneratorData>(requirements);
        nz_handle h = 0;
        if (auto *b = dynamic_cast<GeneratorDataBatch *>(d)) {
            check(nz_fractal_batch(ctx, (int)noiseType, b->data->ptr, b->resolution, b->count, b->positions, hurst,
                                   startingAmplitude, stepdown, detuneRate, octaves, noiseSize, dependency.id, &h),
                  "nz_fractal_batch");
            jobHandle = done(h);
            return;
        }
        check(nz_fractal(ctx, (int)noiseType, d->data->ptr, d->resolution, hurst, startingAmplitude, stepdown,
                         detuneRate, octaves, d->xpos, d->zpos, noiseSize, dependency.id, &h), "nz_fractal");
        jobHandle = done(h);
    }
};

class TmpStage : public PipelineStage {  // stages that own one scratch plane
  public:
    using PipelineStage::PipelineStage;
    void ResizeNativeContainers(size_t) override { tmp.reset(new DeviceTile(ctx, dataLength)); }
    void OnDestroy() override { tmp.reset(); }

  protected:
    std::unique_ptr<DeviceTile> tmp;
};

class KernelFilterStage : public TmpStage {
  public:
    using TmpStage::TmpStage;
    int filter = NZ_GAUSS9_S1;
    int iterations = 1;
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = CheckRequirements<GeneratorData>(requirements);
        nz_handle h = 0;
        if (d->write && filter != NZ_SOBEL3_2D) {
            nz_rw_tile t = rw_pair(d, tile_count(d));
            check(nz_kernel_filter_stage_rw(ctx, &t, filter, iterations, dependency.id, &h), "nz_kernel_filter_stage_rw");
            rw_adopt(d, t);
            jobHandle = done(h);
            return;
        }
        if (auto *b = dynamic_cast<GeneratorDataBatch *>(d)) {
            check(nz_kernel_filter_stage_batch(ctx, b->data->ptr, tmp->ptr, filter, iterations, b->resolution, b->count,
                                               dependency.id, &h), "nz_kernel_filter_stage_batch");
            jobHandle = done(h);
            return;
        }
        check(nz_kernel_filter_stage(ctx, d->data->ptr, tmp->ptr, filter, iterations, d->resolution, dependency.id, &h),
              "nz_kernel_filter_stage");
        jobHandle = done(h);
    }
};

inline int limitWidth(int width) {  // BlurHelper.limitWidth, Filter/Kernel/Blur/BlurKernels.cs:29-36
    if (width % 2 == 0) width += 1;
    if (width > 25) width = 25;
    return width < 3 ? 3 : width;
}

class StageGaussianBlur : public TmpStage {
  public:
    using TmpStage::TmpStage;
    int iterations = 1, sigma = 0, width = 3;
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = CheckRequirements<GeneratorData>(requirements);
        nz_handle h = 0;
        if (d->write) {
            nz_rw_tile t = rw_pair(d, tile_count(d));
            check(nz_gauss_blur_stage_rw(ctx, &t, limitWidth(width), sigma, iterations, dependency.id, &h),
                  "nz_gauss_blur_stage_rw");
            rw_adopt(d, t);
            jobHandle = done(h);
            return;
        }
        if (auto *b = dynamic_cast<GeneratorDataBatch *>(d)) {
            check(nz_gauss_blur_stage_batch(ctx, b->data->ptr, tmp->ptr, limitWidth(width), sigma, iterations, b->resolution,
                                            b->count, dependency.id, &h), "nz_gauss_blur_stage_batch");
            jobHandle = done(h);
            return;
        }
        check(nz_gauss_blur_stage(ctx, d->data->ptr, tmp->ptr, limitWidth(width), sigma, iterations, d->resolution,
                                  dependency.id, &h), "nz_gauss_blur_stage");
        jobHandle = done(h);
    }
};

class StageSmoothBlur : public TmpStage {
  public:
    using TmpStage::TmpStage;
    int iterations = 1, width = 1;
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = CheckRequirements<GeneratorData>(requirements);
        nz_handle h = 0;
        if (d->write && tile_count(d) == 1) {
            nz_rw_tile t = rw_pair(d, 1);
            check(nz_smooth_blur_stage_rw(ctx, &t, limitWidth(width), iterations, dependency.id, &h),
                  "nz_smooth_blur_stage_rw");
            rw_adopt(d, t);
            jobHandle = done(h);
            return;
        }
        check(nz_smooth_blur_stage(ctx, d->data->ptr, tmp->ptr, limitWidth(width), iterations, d->resolution,
                                   dependency.id, &h), "nz_smooth_blur_stage");
        jobHandle = done(h);
    }
};

class ErosionStage : public TmpStage {  // ErosionKernelJob x iterations (no stage wrapper in the reference)
  public:
    using TmpStage::TmpStage;
    int iterations = 1;
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = CheckRequirements<GeneratorData>(requirements);
        nz_handle h = 0;
        if (d->write) {
            nz_rw_tile t = rw_pair(d, tile_count(d));
            check(nz_erosion_stage_rw(ctx, &t, iterations, dependency.id, &h), "nz_erosion_stage_rw");
            rw_adopt(d, t);
            jobHandle = done(h);
            return;
        }
        if (auto *b = dynamic_cast<GeneratorDataBatch *>(d)) {
            check(nz_erosion_stage_batch(ctx, b->data->ptr, tmp->ptr, iterations, b->resolution, b->count, dependency.id,
                                         &h), "nz_erosion_stage_batch");
            jobHandle = done(h);
            return;
        }
        check(nz_erosion_stage(ctx, d->data->ptr, tmp->ptr, iterations, d->resolution, dependency.id, &h),
              "nz_erosion_stage");
        jobHandle = done(h);
    }
};

// ---- element-wise stages (Filter/ConstantStage.cs, Filter/Reduce/ReduceStage.cs, Filter/Curve/CurveStage.cs,
//      Filter/Kernel/Blur/StageThermalErosion.cs) ---------------------------------------------------------
struct ReduceData : StageIO {  // StageIOTypes/ReduceData.cs
    int resolution = 512, xpos = 0, zpos = 0;
    DeviceTile *rightData = nullptr;
};

class ConstantStage : public TmpStage {
  public:
    using TmpStage::TmpStage;
    int operation = 0;  // ConstantOperationType {MULTIPLY, BINARIZE}
    float value = 0.5f;
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = CheckRequirements<GeneratorData>(requirements);
        nz_handle h = 0;
        check(nz_constant_job(ctx, operation, d->data->ptr, tmp->ptr, value, d->resolution, dependency.id, &h),
              "nz_constant_job");
        jobHandle = done(h);
    }
};

class ReduceStage : public TmpStage {
  public:
    using TmpStage::TmpStage;
    int operation = 0;  // ReductionType {SUBTRACT, MULTIPLY, ROOTSUMSQUARES, MAX, MIN}
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = CheckRequirements<ReduceData>(requirements);
        nz_handle h = 0;
        check(nz_reduction_job(ctx, operation, d->data->ptr, d->rightData->ptr, tmp->ptr, d->resolution, dependency.id, &h),
              "nz_reduction_job");
        jobHandle = done(h);
    }
    void TransformData(PipelineWorkItem &inputData) override {  // ReduceStage.cs:53-62
        auto *d = static_cast<ReduceData *>(inputData.data);
        out.uuid = d->uuid;
        out.data = d->data;
        out.resolution = d->resolution;
        out.xpos = d->xpos;
        out.zpos = d->zpos;
        inputData.data = &out;
    }

  private:
    GeneratorData out;
};

class CurveStage : public TmpStage {
  public:
    using TmpStage::TmpStage;
    std::function<float(float)> unityCurve = [](float t) { return t; };  // AnimationCurve.Evaluate
    int samples = 256;
    void ResizeNativeContainers(size_t n) override {
        TmpStage::ResizeNativeContainers(n);
        std::vector<float> host(samples);
        for (int i = 0; i < samples; i++) host[i] = unityCurve((float)i / (float)samples);  // ExtractCurve :32-40
        curve.reset(new DeviceTile(ctx, samples));
        curve->CopyFrom(host.data());
    }
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = CheckRequirements<GeneratorData>(requirements);
        nz_handle h = 0;
        check(nz_curve_job(ctx, d->data->ptr, tmp->ptr, curve->ptr, samples, d->resolution, dependency.id, &h),
              "nz_curve_job");
        jobHandle = done(h);
    }
    void OnDestroy() override {
        TmpStage::OnDestroy();
        curve.reset();
    }

  private:
    std::unique_ptr<DeviceTile> curve;
};

class StageThermalErosion : public PipelineStage {
  public:
    using PipelineStage::PipelineStage;
    int iterations = 1, talus = 45;
    float increment = 0.5f, meshHeightWidthRatio = 0.75f;
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = CheckRequirements<GeneratorData>(requirements);
        nz_handle h = 0;
        check(nz_thermal_erosion(ctx, d->data->ptr, (float)talus, increment, meshHeightWidthRatio, iterations,
                                 d->resolution, dependency.id, &h), "nz_thermal_erosion");
        jobHandle = done(h);
    }
};

class FlowMapStage : public PipelineStage {
  public:
    using PipelineStage::PipelineStage;
    int iterations = 5;
    float normMin = -.1f, normMax = .1f;
    void ResizeNativeContainers(size_t) override {
        work.reset(new DeviceTile(ctx, 11 * dataLength));  // nz_flowmap_stage_work_floats per tile of the payload
    }
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *g = dynamic_cast<GeneratorData *>(requirements.data);
        if (!g) throw std::runtime_error("Unhandled stageio");
        resolution = g->resolution;
        auto *d = CheckRequirements<GeneratorData>(requirements);
        nz_handle h = 0;
        if (d->write) {
            nz_rw_tile t = rw_pair(d, tile_count(d));
            check(nz_flowmap_stage_rw(ctx, &t, work->ptr, iterations, normMin, normMax, dependency.id, &h),
                  "nz_flowmap_stage_rw");
            rw_adopt(d, t);
            jobHandle = done(h);
            return;
        }
        if (auto *b = dynamic_cast<GeneratorDataBatch *>(d)) {
            check(nz_flowmap_stage_batch(ctx, b->data->ptr, work->ptr, iterations, normMin, normMax, b->resolution,
                                         b->count, dependency.id, &h), "nz_flowmap_stage_batch");
            jobHandle = done(h);
            return;
        }
        check(nz_flowmap_stage(ctx, d->data->ptr, work->ptr, iterations, normMin, normMax, d->resolution,
                               dependency.id, &h), "nz_flowmap_stage");
        jobHandle = done(h);
    }
    void OnDestroy() override { work.reset(); }

  private:
    int resolution = 0;
    std::unique_ptr<DeviceTile> work;
};

class MeshTileStage : public PipelineStage {
  public:
    using PipelineStage::PipelineStage;
    int meshType = NZ_MESH_SQUARE_GRID;
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = dynamic_cast<MeshStageData *>(requirements.data);
        if (!d || !d->mesh) throw std::runtime_error("Unhandled stageio");
        MeshBuffers &m = *d->mesh;
        size_t nv = nz_mesh_vertex_count(d->resolution) * d->count, ni = nz_mesh_index_count(d->resolution) * d->count;
        if (m.vertexCount != nv) {  // Mesh.AllocateWritableMeshData(1)
            m.vertices.reset(new DeviceTile(ctx, nv * 12));
            m.indices.reset(new DeviceTile(ctx, ni));
            m.vertexCount = nv;
            m.indexCount = ni;
        }
        nz_handle h = 0;
        if (d->count > 1) {
            check(nz_heightmap_mesh_batch(ctx, meshType, m.vertices->ptr, reinterpret_cast<uint32_t *>(m.indices->ptr),
                                          d->resolution, d->inputResolution, d->marginPix, d->tileHeight, d->tileSize,
                                          d->data->ptr, d->count, dependency.id, &h), "nz_heightmap_mesh_batch");
            jobHandle = done(h);
            return;
        }
        check(nz_heightmap_mesh(ctx, meshType, m.vertices->ptr, reinterpret_cast<uint32_t *>(m.indices->ptr),
                                d->resolution, d->inputResolution, d->marginPix, d->tileHeight, d->tileSize,
                                d->data->ptr, dependency.id, &h), "nz_heightmap_mesh");
        jobHandle = done(h);
    }
};

// ---- CropStage (Filter/Sample/CropStage.cs:11-19; the job's offset stays 0 as in CropJob.cs:43-59) -----------
class CropStage : public PipelineStage {
  public:
    using PipelineStage::PipelineStage;
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = dynamic_cast<DownsampleData *>(requirements.data);
        if (!d) throw std::runtime_error("Unhandled stageio");
        nz_handle h = 0;
        check(nz_crop_job(ctx, d->inputData->ptr, d->inputResolution, d->data->ptr, d->resolution, dependency.id, &h),
              "nz_crop_job");
        jobHandle = done(h);
    }
};

// ---- context stages (Pipeline/PipelineState/Stage/*.cs, Mesh/Stage/MeshTileReferenceDataStage.cs) -----------
inline std::string contextBufferName(int xpos, int zpos, int resolution, const std::string &alias) {
    return std::to_string(xpos) + "_" + std::to_string(zpos) + "__" + std::to_string(resolution) + "__" + alias;
}

class ReadGeneratorContextStage : public PipelineStage {  // ReadGeneratorContextStage.cs:13-46
  public:
    using PipelineStage::PipelineStage;
    std::string contextAlias;
    bool IsSchedulable(const PipelineWorkItem &job) override {
        auto *gd = dynamic_cast<GeneratorData *>(job.data);
        if (!job.stageManager || !gd) return false;
        std::string name = contextBufferName(gd->xpos, gd->zpos, gd->resolution, contextAlias);
        return job.stageManager->BufferExists(name) && !job.stageManager->IsLocked(name);
    }
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *gd = CheckRequirements<GeneratorData>(requirements);
        size_t res = (size_t)gd->resolution * gd->resolution;
        DeviceTile *buffer = requirements.stageManager->GetBuffer(
            contextBufferName(gd->xpos, gd->zpos, gd->resolution, contextAlias), (long)res);
        nz_handle h = 0;
        check(nz_flush_write_slice(ctx, gd->data->ptr, buffer->ptr, res, dependency.id, &h), "nz_flush_write_slice");
        jobHandle = done(h);
    }
};

class WriteGeneratorContextStage : public PipelineStage {  // WriteGeneratorContextStage.cs:13-46
  public:
    using PipelineStage::PipelineStage;
    std::string contextAlias;
    bool IsSchedulable(const PipelineWorkItem &job) override {
        auto *gd = dynamic_cast<GeneratorData *>(job.data);
        if (!job.stageManager || !gd) return false;
        return !job.stageManager->IsLocked(contextBufferName(gd->xpos, gd->zpos, gd->resolution, contextAlias));
    }
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *gd = CheckRequirements<GeneratorData>(requirements);
        size_t res = (size_t)gd->resolution * gd->resolution;
        std::string name = contextBufferName(gd->xpos, gd->zpos, gd->resolution, contextAlias);
        DeviceTile *buffer = requirements.stageManager->GetBuffer(name, (long)res);
        nz_handle h1 = 0, h2 = 0;
        check(nz_flush_write_slice(ctx, buffer->ptr, gd->data->ptr, res, dependency.id, &h1), "nz_flush_write_slice");
        check(nz_handle_record(ctx, &h2), "nz_handle_record");  // LockJob: a no-op marker after the copy
        jobHandle = done(h2);
        requirements.stageManager->TrySetLock(name, done(h1), jobHandle);
    }
};

class MeshTileReferenceDataStage : public MeshTileStage {  // Mesh/Stage/MeshTileReferenceDataStage.cs:22-84
  public:
    using MeshTileStage::MeshTileStage;
    std::string contextAlias;
    bool IsSchedulable(const PipelineWorkItem &job) override {
        auto *d = dynamic_cast<MeshStageData *>(job.data);
        if (!job.stageManager || !d) return false;
        std::string name = contextBufferName(d->xpos, d->zpos, d->inputResolution, contextAlias);
        return job.stageManager->BufferExists(name) && !job.stageManager->IsLocked(name);
    }
    void Schedule(PipelineWorkItem &requirements, JobHandle dependency) override {
        auto *d = dynamic_cast<MeshStageData *>(requirements.data);
        if (!d) throw std::runtime_error("Unhandled stageio");
        DeviceTile *own = d->data;
        d->data = requirements.stageManager->GetBuffer(contextBufferName(d->xpos, d->zpos, d->inputResolution, contextAlias),
                                                       (long)d->inputResolution * d->inputResolution);
        try {
            MeshTileStage::Schedule(requirements, dependency);  // the mesh job reads the context buffer (:62-64)
        } catch (...) {
            d->data = own;
            throw;
        }
        d->data = own;
    }
};

// ---- BasePipeline (Pipeline/Executable/Pipeline.cs) --------------------------------------------
class BasePipeline {
  public:
    std::string alias = "Unnamed Pipeline";
    PipelineStateManager *contextManager = nullptr;  // handed to every work item (Pipeline.cs:76-104)
    explicit BasePipeline(std::vector<PipelineStage *> stages) : stage_instances(std::move(stages)) { Setup(); }
    virtual ~BasePipeline() = default;
    virtual std::vector<BasePipeline *> GetDependencies() { return {this}; }  // Pipeline.cs:63-65
    bool Idle() const { return queue.empty() && dependencyHell.empty() && !pipelineRunning && !pipelineBeingScheduled; }
    size_t Parked() const { return dependencyHell.size(); }

    void Enqueue(StageIO *input, std::function<void(StageIO *, JobHandle)> scheduleAction = nullptr,
                 std::function<void(StageIO *)> completeAction = nullptr, JobHandle dependency = JobHandle()) {
        queue.push_back(PipelineWorkItem{input, completeAction, scheduleAction, dependency, contextManager});
    }
    void Schedule(PipelineWorkItem wi) {
        activeItem = std::move(wi);
        if (stage_instances.empty()) throw std::runtime_error("No stages in pipeline");
        pipelineBeingScheduled = true;
        stage_instances[0]->ReceiveHandledInput(activeItem, activeItem.dependency);
    }
    bool WorkIsSchedulable(const PipelineWorkItem &item) {  // Pipeline.cs:256-265
        bool ready = true;
        for (auto *s : stage_instances) ready = s->IsSchedulable(item) && ready;
        return ready;
    }
    bool GetNextJob(PipelineWorkItem &out) {  // Pipeline.cs:183-214: parked items first, then the queue
        for (size_t i = 0; i < dependencyHell.size(); i++)
            if (WorkIsSchedulable(dependencyHell[i])) {
                out = dependencyHell[i];
                dependencyHell.erase(dependencyHell.begin() + i);
                return true;
            }
        while (!queue.empty()) {
            PipelineWorkItem wi = queue.front();
            queue.pop_front();
            if (WorkIsSchedulable(wi)) {
                out = wi;
                return true;
            }
            dependencyHell.push_back(wi);
        }
        return false;
    }
    virtual void Update() {
        if (!pipelineRunning && !pipelineBeingScheduled) {
            PipelineWorkItem wi;
            if (GetNextJob(wi)) Schedule(wi);
        }
    }
    // Nobody outside this pipeline has been handed a handle of the pass that is running: the work item carries no
    // scheduledAction and no stage has a scheduled-action hook beyond its hand-over to the next stage (a joint, a downstream
    // pipeline).  Work scheduled on such a handle has consumed the failed pass's planes and cannot be recalled from here.
    bool RetryIsLocal() const {
        if (activeItem.scheduledAction) return false;
        for (size_t i = 0; i < stage_instances.size(); i++)
            if (stage_instances[i]->OnStageScheduledAction.size() != 1) return false;  // (Setup() gave every stage exactly one)
        return true;
    }
    // pipelineHandle.Complete().  NZ_ERR_RETRY -- a chained kernel-filter launch timed out, the planes computed since are
    // invalid and the context has switched to separate launches -- is answered once by running the work item again, when
    // the pipeline regenerates its tile from scratch (its first stage is the NoiseStage) and the failed pass is this
    // pipeline's own business (RetryIsLocal).  The failed pass is wound up first (OnStageComplete, as after any pass); the
    // item's dependency was satisfied by the first pass and is not applied again.  Otherwise the error goes to the caller.
    void CompleteActive() {
        try {
            pipelineHandle.Complete();
        } catch (const NoizeError &e) {
            if (e.status != NZ_ERR_RETRY || !RegeneratesItsTile() || !RetryIsLocal()) throw;
            for (auto *s : stage_instances) s->OnStageComplete();
            pipelineRunning = false;
            activeItem.dependency = JobHandle();
            Schedule(activeItem);
            pipelineHandle.Complete();  // (a second failure is the caller's)
        }
    }
    bool LateUpdate() {
        if (pipelineRunning && pipelineHandle.IsCompleted()) {
            CompleteActive();
            for (auto *s : stage_instances) s->OnStageComplete();
            if (activeItem.completeAction) activeItem.completeAction(activeItem.data);
            pipelineRunning = false;
            return true;
        }
        return false;
    }
    void RunToCompletion() {  // until the queue is drained or only unschedulable items are left
        while (!queue.empty() || !dependencyHell.empty() || pipelineRunning) {
            Update();
            if (pipelineRunning) {
                CompleteActive();
                LateUpdate();
            } else if (queue.empty()) {
                break;  // everything left is parked on a dependency another pipeline has to satisfy
            }
        }
    }
    virtual void Destroy() {
        for (auto *s : stage_instances) s->OnDestroy();
    }

  protected:
    bool RegeneratesItsTile() const;  // the first stage is the NoiseStage
    void Setup() {
        PipelineStage *previous = nullptr;
        for (auto *stage : stage_instances) {
            if (previous)
                previous->OnStageScheduledAction.push_back(
                    [stage](PipelineWorkItem &wi, JobHandle h) { stage->ReceiveHandledInput(wi, h); });
            previous = stage;
        }
        if (previous)
            previous->OnStageScheduledAction.push_back([this](PipelineWorkItem &wi, JobHandle h) {
                pipelineHandle = h;
                pipelineRunning = true;
                pipelineBeingScheduled = false;
                if (activeItem.scheduledAction) activeItem.scheduledAction(wi.data, h);
            });
    }
    std::vector<PipelineStage *> stage_instances;
    std::deque<PipelineWorkItem> queue;
    std::vector<PipelineWorkItem> dependencyHell;
    PipelineWorkItem activeItem;
    JobHandle pipelineHandle;
    bool pipelineBeingScheduled = false, pipelineRunning = false;
};

// ---- ReducePipeline (Pipeline/Executable/ReducePipeline.cs:31-166) ------------------------------
// One work item -> the same tile requested from both upstream pipelines (the right one into a plane this
// pipeline owns) -> this pipeline's stages on a ReduceData of the two planes.
class ReducePipeline : public BasePipeline {
  public:
    ReducePipeline(nz_ctx *ctx, std::vector<PipelineStage *> stages, BasePipeline *left, BasePipeline *right)
        : BasePipeline(std::move(stages)), ctx(ctx), upstreamPipelineLeft(left), upstreamPipelineRight(right) {}

    std::vector<BasePipeline *> GetDependencies() override {  // :52-62
        std::vector<BasePipeline *> up{upstreamPipelineLeft, upstreamPipelineRight, this};
        for (auto *p : upstreamPipelineLeft->GetDependencies()) up.push_back(p);
        for (auto *p : upstreamPipelineRight->GetDependencies()) up.push_back(p);
        return up;
    }
    void Update() override {  // OnUpdate :64-80
        if (!pipelineRunning && !pipelineBeingScheduled && !upstreamsRunning && !queue.empty()) {
            PipelineWorkItem wi = queue.front();
            queue.pop_front();
            upstreamsRunning = true;
            ScheduleUpstreams(wi);
        }
    }
    void RunToCompletion() {  // the frame loop over this pipeline and everything upstream of it
        std::vector<BasePipeline *> pipes;
        for (auto *p : GetDependencies())
            if (std::find(pipes.begin(), pipes.end(), p) == pipes.end()) pipes.push_back(p);
        bool busy = true;
        while (busy) {
            for (auto *p : pipes) p->Update();
            for (auto *p : pipes) p->LateUpdate();
            busy = upstreamsRunning;
            for (auto *p : pipes) busy = busy || !p->Idle();
        }
    }
    void Destroy() override {  // :157-163
        rightData.reset();
        BasePipeline::Destroy();
    }

  private:
    void ScheduleUpstreams(const PipelineWorkItem &wi) {  // :82-121
        auto *leftData = dynamic_cast<GeneratorData *>(wi.data);
        if (!leftData) throw std::runtime_error("Unhandled stageio");
        if (!rightData || rightData->Length != leftData->data->Length)
            rightData.reset(new DeviceTile(ctx, leftData->data->Length));
        left = leftData;
        rightIO = *leftData;
        rightIO.data = rightData.get();
        doneLeft = doneRight = false;
        action = wi.completeAction;
        upstreamPipelineLeft->Enqueue(left, nullptr, [this](StageIO *res) { OnCompleteUpstream(res, true); });
        upstreamPipelineRight->Enqueue(&rightIO, nullptr, [this](StageIO *res) { OnCompleteUpstream(res, false); });
    }
    void OnCompleteUpstream(StageIO *res, bool isLeft) {  // :123-149
        (isLeft ? doneLeft : doneRight) = true;
        if (!(doneLeft && doneRight)) return;
        upstreamsRunning = false;
        auto *d = static_cast<GeneratorData *>(res);
        joined.uuid = d->uuid;
        joined.resolution = d->resolution;
        joined.data = left->data;
        joined.rightData = rightIO.data;
        joined.xpos = d->xpos;
        joined.zpos = d->zpos;
        Schedule(PipelineWorkItem{&joined, action, nullptr, JobHandle(), contextManager});
    }
    nz_ctx *ctx;
    BasePipeline *upstreamPipelineLeft, *upstreamPipelineRight;
    bool upstreamsRunning = false, doneLeft = false, doneRight = false;
    std::unique_ptr<DeviceTile> rightData;
    GeneratorData *left = nullptr;
    GeneratorData rightIO;
    ReduceData joined;
    std::function<void(StageIO *)> action;
};

// ---- live erosion (BASELINE config 4): Component/LiveErosion.cs:200-436 without the MonoBehaviour ----------------
enum class ErosionMode { ALL_EROSION, ONLY_THERMAL_EROSION, THERMAL_FLOW_WATER, ONLY_FLOW_WATER };  // LiveErosionDataTypes.cs:28-33

struct ErosionSettings {  // ScriptableObject/ErosionSettings.cs:5-124 (defaults = Reset())
    int CYCLES = 3, PARTICLES_PER_CYCLE = 1000;
    ErosionMode BEHAVIOR = ErosionMode::ALL_EROSION;
    float INERTIA = 0.5f, GRAVITY = 1.0f, DRAG = 0.001f, FRICTION = 0.01f, EVAP = 0.01f, EROSION = 1.0f, DEPOSITION = 0.1f;
    float FLOW_HEIGHT_CONTRIBUTION = 25.0f, SLOW_CULL_ANGLE = 3.0f, SLOW_CULL_SPEED = 0.11f, CAPACITY = 3.0f;
    int MAXAGE = 100, WATER_STEPS = 10;
    float SURFACE_EVAPORATION_RATE = 0.1f, POOL_PLACEMENT_MULTIPLIER = 0.5f, TRACK_PLACEMENT_MULTIPLIER = 80.0f,
          FLOW_LOSS_RATE = 0.05f;
    int PILING_RADIUS = 15;
    float MIN_PILE_INCREMENT = 1.0f, PILE_THRESHOLD = 2.0f;
    bool ENABLE_THERMAL = true;
    float TALUS = 55.0f, THERMAL_STEP = 0.6f;
    int THERMAL_CYCLES = 1;

    nz_erosion_params AsParameters() const {  // :96-123
        nz_erosion_params ep{};
        ep.INERTIA = INERTIA; ep.GRAVITY = GRAVITY; ep.DRAG = DRAG; ep.FRICTION = FRICTION; ep.EVAP = EVAP;
        ep.EROSION = EROSION; ep.DEPOSITION = DEPOSITION; ep.FLOW_HEIGHT_CONTRIBUTION = FLOW_HEIGHT_CONTRIBUTION;
        ep.SLOW_CULL_ANGLE = SLOW_CULL_ANGLE; ep.SLOW_CULL_SPEED = SLOW_CULL_SPEED;
        ep.CAPACITY = BEHAVIOR == ErosionMode::ALL_EROSION ? CAPACITY : 0.0f;
        ep.MAXAGE = MAXAGE;
        ep.TERMINAL_VELOCITY = 1.0f / DRAG;
        ep.SURFACE_EVAPORATION_RATE = SURFACE_EVAPORATION_RATE;
        ep.POOL_PLACEMENT_MULTIPLIER = BEHAVIOR == ErosionMode::ONLY_THERMAL_EROSION ? 0.0f : POOL_PLACEMENT_MULTIPLIER;
        ep.TRACK_PLACEMENT_MULTIPLIER = TRACK_PLACEMENT_MULTIPLIER; ep.FLOW_LOSS_RATE = FLOW_LOSS_RATE;
        ep.PILING_RADIUS = PILING_RADIUS; ep.MIN_PILE_INCREMENT = MIN_PILE_INCREMENT; ep.PILE_THRESHOLD = PILE_THRESHOLD;
        return ep;
    }
};

// Owns poolMap / streamMap / particleTrack (planes indexed x * res + z), the particle queue and the erosive events, and
// chains one Update's worth of jobs on the context's stream.  `seeds`: one per cycle (the reference draws them from
// UnityEngine.Random, MultiThreadErosionJob.cs:50): the same seeds give the same planes on every run.
class LiveErosion {
  public:
    LiveErosion(nz_ctx *c, DeviceTile *height, const nz_tile_set_meta &tm, const ErosionSettings &es, bool perform = true)
        : ctx(c), heightMap(height), tileMeta(tm), erosionSettings(es), performErosion(perform), res(tm.GENERATOR_RES[0]),
          poolMap(c, (size_t)tm.GENERATOR_RES[0] * tm.GENERATOR_RES[0]), streamMap(c, poolMap.Length),
          particleTrack(c, poolMap.Length) {
        if (height->Length != poolMap.Length) throw std::runtime_error("heightMap is not GENERATOR_RES^2 cells");
        std::vector<float> zeros(poolMap.Length, 0.0f);
        poolMap.CopyFrom(zeros.data());
        streamMap.CopyFrom(zeros.data());
        particleTrack.CopyFrom(zeros.data());
        QUEUE_SIZE = perform ? es.PARTICLES_PER_CYCLE : 1;  // :215-219
        const int n = res * res;
        check(nz_particle_queue_create(ctx, std::max(std::max(4 * QUEUE_SIZE, QUEUE_SIZE + n / 8), 1024), &particleQueue),
              "nz_particle_queue_create");
        check(nz_erosive_events_create(ctx, res, &events), "nz_erosive_events_create");
    }
    LiveErosion(const LiveErosion &) = delete;
    // nz_ctx_set_pile_safe: ErodeHeightMaps keeps a copy of the height plane, waits for the pile solver's one-launch form and runs
    // itself again colour by colour should a block of it ever give up (a property of the context)
    void SetSafe(bool on) { check(nz_ctx_set_pile_safe(ctx, on ? 1 : 0), "nz_ctx_set_pile_safe"); }
    int PileRetries() const { return nz_ctx_pile_retries(ctx); }
    ~LiveErosion() {
        jobHandle.Complete();
        if (particleQueue) nz_particle_queue_destroy(ctx, particleQueue);
        if (events) nz_erosive_events_destroy(ctx, events);
    }

    JobHandle TriggerQueuedBeyerMT(const std::vector<int> &seeds) {  // :378-436
        const ErosionSettings &es = erosionSettings;
        const nz_erosion_params ep = es.AsParameters();
        const nz_tile_set_meta &tm = tileMeta;
        nz_handle h = 0;
        // The jobs of a cycle are links of ONE chain on the context's stream: with fewHandles only the handles somebody waits
        // for are asked of the library (out = NULL otherwise) -- a handle is an event record, ~3 us of the stream
        const bool all = !fewHandles;
        auto link = [&](bool wanted) -> nz_handle * { return wanted || all ? &h : (h = 0, (nz_handle *)nullptr); };
        if (performErosion) {
            if ((int)seeds.size() < es.CYCLES) throw std::runtime_error("one seed per cycle");
            for (int i = 0; i < es.CYCLES; i++) {
                const bool last = i + 1 == es.CYCLES;  // the chain's last handle is the component's jobHandle
                nz_handle dep = h;
                if (es.ENABLE_THERMAL && es.BEHAVIOR != ErosionMode::ONLY_FLOW_WATER) {  // `TILE_SIZE.x / HEIGHT`: int / int (:386)
                    check(nz_thermal_erosion(ctx, heightMap->ptr, es.TALUS, es.THERMAL_STEP, (float)(tm.TILE_SIZE[0] / tm.HEIGHT),
                                             es.THERMAL_CYCLES, res, dep, link(false)), "nz_thermal_erosion");
                    dep = h;
                }
                if (es.BEHAVIOR != ErosionMode::ONLY_FLOW_WATER) {
                    check(nz_fill_beyer_queue(ctx, particleQueue, &ep, &tm, particleGenerationID % 4, res, QUEUE_SIZE, seeds[i],
                                              std::min(10, QUEUE_SIZE), dep, link(false)), "nz_fill_beyer_queue");
                    dep = h;
                }
                check(nz_queued_beyer_cycle(ctx, heightMap->ptr, poolMap.ptr, streamMap.ptr, particleTrack.ptr, particleQueue, events,
                                            &ep, &tm, EVENT_LIMIT, res, dep, link(false)), "nz_queued_beyer_cycle");
                dep = h;
                check(nz_process_beyer_erosive_events(ctx, heightMap->ptr, poolMap.ptr, streamMap.ptr, particleTrack.ptr, events, &ep,
                                                      &tm, res, dep, link(false)), "nz_process_beyer_erosive_events");
                // CombineDependencies(ClearQueueJob, ErodeHeightMaps, UpdateFlowFromTrackJob), all behind the event reduction
                // (:408-412): the clear, then the two siblings as one call (the pile solver's launch carries the flow update's
                // workgroups); fuseSiblings = false: the two entries one after the other on the context's stream
                dep = h;
                check(nz_clear_particle_queue(ctx, particleQueue, dep, link(false)), "nz_clear_particle_queue");
                dep = h;
                if (fuseSiblings) {
                    check(nz_erode_height_maps_and_flow(ctx, heightMap->ptr, events, poolMap.ptr, streamMap.ptr, particleTrack.ptr, &ep,
                                                        &tm, res, dep, link(false)), "nz_erode_height_maps_and_flow");
                } else {
                    check(nz_erode_height_maps(ctx, heightMap->ptr, events, &ep, &tm, res, dep, link(false)), "nz_erode_height_maps");
                    dep = h;
                    check(nz_update_flow_from_track(ctx, poolMap.ptr, streamMap.ptr, particleTrack.ptr, ep.FLOW_LOSS_RATE,
                                                    ep.SURFACE_EVAPORATION_RATE, (float)tm.HEIGHT, res, dep, link(false)), "nz_update_flow_from_track");
                }
                dep = h;
                check(nz_pool_automata_job(ctx, poolMap.ptr, heightMap->ptr, particleQueue, &ep, &tm, es.WATER_STEPS, res,
                                           performErosion ? 1 : 0, dep, link(last)), "nz_pool_automata_job");
            }
        }
        jobHandle = JobHandle{ctx, h};
        particleGenerationID += 1;  // Update() :341
        return jobHandle;
    }

    nz_ctx *ctx;
    bool fewHandles = true;         // false: a handle out of every job, as the reference schedules them (one event record each)
    bool fuseSiblings = true;       // ErodeHeightMaps + UpdateFlowFromTrackJob as one call (nz_erode_height_maps_and_flow)
    DeviceTile *heightMap;
    nz_tile_set_meta tileMeta;
    ErosionSettings erosionSettings;
    bool performErosion;
    int res, QUEUE_SIZE = 0, particleGenerationID = 0, EVENT_LIMIT = 1500;
    DeviceTile poolMap, streamMap, particleTrack;
    nz_particle_queue *particleQueue = nullptr;
    nz_erosive_events *events = nullptr;
    JobHandle jobHandle;
};


// The stock stage list NoiseStage -> [KernelFilterStage] -> [FlowMapStage] -> [ErosionStage] (README.md:23-32), all on one
// context, as nz_terrain_params (what ShardedPipeline hands to nz_sharded_create); false if the list is anything else.
inline bool stockListParams(const std::vector<PipelineStage *> &stages, nz_terrain_params *tp, NoiseStage **noise) {
    if (stages.empty()) return false;
    auto *n = dynamic_cast<NoiseStage *>(stages[0]);
    if (!n) return false;
    KernelFilterStage *f = nullptr;
    FlowMapStage *w = nullptr;
    ErosionStage *e = nullptr;
    int k = 0;  // the optional stages must come in this order, each at most once
    for (size_t i = 1; i < stages.size(); i++) {
        PipelineStage *s = stages[i];
        if (s->Context() != n->Context()) return false;
        if (k < 1 && typeid(*s) == typeid(KernelFilterStage)) { f = static_cast<KernelFilterStage *>(s); k = 1; }
        else if (k < 2 && typeid(*s) == typeid(FlowMapStage)) { w = static_cast<FlowMapStage *>(s); k = 2; }
        else if (k < 3 && typeid(*s) == typeid(ErosionStage)) { e = static_cast<ErosionStage *>(s); k = 3; }
        else return false;
    }
    if (typeid(*n) != typeid(NoiseStage) || (f && f->filter == NZ_SOBEL3_2D)) return false;
    *tp = nz_terrain_params{};
    tp->noiseType = (int)n->noiseType;
    tp->hurst = n->hurst; tp->startingAmplitude = n->startingAmplitude; tp->stepdown = n->stepdown; tp->detuneRate = n->detuneRate;
    tp->octaves = n->octaves; tp->noiseSize = n->noiseSize;
    tp->filter = f ? f->filter : 0; tp->filterIterations = f ? f->iterations : 0;
    tp->flowIterations = w ? w->iterations : 0; tp->normMin = w ? w->normMin : 0.f; tp->normMax = w ? w->normMax : 0.f;
    tp->erosionIterations = e ? e->iterations : 0;
    if (noise) *noise = n;
    return true;
}

inline bool BasePipeline::RegeneratesItsTile() const {
    return !stage_instances.empty() && typeid(*stage_instances[0]) == typeid(NoiseStage);
}

// ---- one large grid over the GPUs of a node (new-framework feature; include/noize_hip.h, nz_comm.cpp) -------------------
// The reference's host asks for one tile at a time (BasePipeline.Schedule, Pipeline/Executable/Pipeline.cs:104-128;
// Scripts/MeshTileGenerator.cs:166-192); a grid larger than a tile is cut into row stripes, one process per GPU, and the
// stock stage list runs on all of them with RCCL neighbour halo exchange.
class Comm {  // nz_comm: ncclCommInitRank on the context's device + the exchanges' stream
  public:
    static std::array<uint8_t, NZ_COMM_ID_BYTES> UniqueId() {  // by ONE rank; the bytes travel out of band
        std::array<uint8_t, NZ_COMM_ID_BYTES> id{};
        check(nz_comm_unique_id(id.data()), "nz_comm_unique_id");
        return id;
    }
    Comm(nz_ctx *ctx, const std::array<uint8_t, NZ_COMM_ID_BYTES> &id, int rank, int world) {
        check(nz_comm_init(ctx, id.data(), rank, world, &h), "nz_comm_init");
    }
    // nz_comm_destroy refuses while ShardedPipelines still hold the communicator (destroy them first -- objects declared
    // after the Comm are, by the language's order).  Close() reports that; the destructor can only say so.
    void Close() {
        if (!h) return;
        check(nz_comm_destroy(h), "nz_comm_destroy");
        h = nullptr;
    }
    ~Comm() {
        if (h && nz_comm_destroy(h) != NZ_OK)
            std::fprintf(stderr, "noize::Comm: communicator leaked (%s)\n", nz_last_error());
    }
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    int Rank() const { return nz_comm_rank(h); }
    int World() const { return nz_comm_world(h); }
    nz_comm *h = nullptr;
};

class ShardedPipeline {
  public:
    // `stages`: the stock list (stockListParams); comm may be null on one rank (device copies instead of RCCL);
    // stripes = 0: one per rank
    ShardedPipeline(nz_ctx *ctx, Comm *comm, const std::vector<PipelineStage *> &stages, int grows, int cols, int stripes = 0,
                    int haloMode = NZ_HALO_EXCHANGE, int overlap = 0, int xpos = 0, int zpos = 0)
        : ctx(ctx) {
        nz_terrain_params tp{};
        if (!stockListParams(stages, &tp, nullptr)) throw std::runtime_error("ShardedPipeline: not the stock stage list");
        nz_sharded_desc d{};
        d.grows = grows;
        d.cols = cols;
        d.stripes = stripes > 0 ? stripes : (comm ? comm->World() : 1);
        d.haloMode = haloMode;
        d.overlap = overlap;
        d.xpos = xpos;
        d.zpos = zpos;
        check(nz_sharded_create(ctx, comm ? comm->h : nullptr, &d, &tp, &h), "nz_sharded_create");
    }
    ~ShardedPipeline() { nz_sharded_destroy(h); }
    ShardedPipeline(const ShardedPipeline &) = delete;
    ShardedPipeline &operator=(const ShardedPipeline &) = delete;
    int LocalStripes() const { return nz_sharded_local_stripes(h); }
    JobHandle Schedule(JobHandle dependency = JobHandle()) {  // one pass over every local stripe (enqueue only)
        nz_handle out = 0;
        check(nz_sharded_pipeline(ctx, h, nullptr, dependency.id, &out), "nz_sharded_pipeline");
        return JobHandle{ctx, out};
    }
    // local stripe i: first owned global row, owned rows, host copy of them (rows x cols floats)
    int OwnedRows(int i, int *firstGlobalRow = nullptr) const {
        nz_stripe st{};
        check(nz_sharded_stripe(h, i, &st, nullptr, nullptr), "nz_sharded_stripe");
        if (firstGlobalRow) *firstGlobalRow = st.grow0 + st.own0;
        return st.own1 - st.own0;
    }
    void CopyOwnedRows(int i, float *host) const {
        nz_stripe st{};
        float *res = nullptr;
        check(nz_sharded_stripe(h, i, &st, nullptr, &res), "nz_sharded_stripe");
        const size_t n = (size_t)(st.own1 - st.own0) * st.cols;
        nz_handle done = 0;
        check(nz_tile_download(ctx, res + (size_t)st.own0 * st.cols, host, n, 0, &done), "nz_tile_download");
        check(nz_handle_wait(ctx, done), "nz_handle_wait");
    }
    nz_ctx *ctx;
    nz_sharded *h = nullptr;
};

}  // namespace noize
