// host_demo.cpp -- the metric pipeline driven from the C++ host mirror (noize_pipeline.hpp).
// Writes the resulting plane as raw little-endian fp32 so the test suite can compare it with the oracle.
//   usage: host_demo <resolution> <out.f32> [gauss_iterations flow_iterations erosion_iterations]
//          host_demo <resolution> <out.f32> reduce      (ReducePipeline: simplex x cellular, MULTIPLY)
//          host_demo <resolution> <out.f32> context     (producer -> context buffer -> consumer, parked until written)
//          host_demo <resolution> <out.f32> batch <n>   (n tiles at xpos = k * resolution through the batched stage bodies)
//          host_demo <resolution> <out.f32> live <particles> <cycles>   (cellular fBm -> LiveErosion, seeds 3, 14, 25, ...:
//                                                                         height, pool, flow planes back to back)
//          host_demo <resolution> <out.f32> sharded <stripes> [mode overlap]   (ShardedPipeline on ONE rank: the grid as
//                            <stripes> row stripes whose ghost rows travel through ncclSend / ncclRecv; mode 0 recompute,
//                            1 exchange, 2 exchange_once)
//          host_demo <resolution> <out.f32> sharded-rank <rank> <world> <idfile> [mode]   (one process per GPU, device =
//                            rank; rank 0 writes the ncclUniqueId to <idfile>; every rank writes its rows to <out.f32>.<rank>)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>

#include "noize_pipeline.hpp"

using namespace noize;

// a pipeline whose driver does not wait for a work item before it schedules the next one (the `tiles` mode: the tile's
// planes are the pipeline's own, and the stream orders the work items)
struct FreeRunningPipeline : BasePipeline {
    using BasePipeline::BasePipeline;
    void Release() { pipelineRunning = pipelineBeingScheduled = false; }
};

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <resolution> <out.f32> [G F E]\n", argv[0]);
        return 2;
    }
    int res = std::atoi(argv[1]);
    const bool reduce = argc > 3 && std::strcmp(argv[3], "reduce") == 0;
    const bool context = argc > 3 && std::strcmp(argv[3], "context") == 0;
    const int batch = argc > 4 && std::strcmp(argv[3], "batch") == 0 ? std::atoi(argv[4]) : 0;
    // `rw`: the tile is a READ / WRITE plane pair and the stencil stages swap it instead of flushing (nz_*_rw)
    const bool live = argc > 5 && std::strcmp(argv[3], "live") == 0;
    const bool rw = argc > 3 && std::strcmp(argv[3], "rw") == 0;
    const bool sharded = argc > 4 && std::strcmp(argv[3], "sharded") == 0;
    const bool sharded_rank = argc > 6 && std::strcmp(argv[3], "sharded-rank") == 0;
    const bool tiles = argc > 5 && std::strcmp(argv[3], "tiles") == 0;  // <res> <out> tiles <pipelines> <tiles>
    if (rw) argc = 3;
    int G = argc > 3 ? std::atoi(argv[3]) : 17, F = argc > 4 ? std::atoi(argv[4]) : 5, E = argc > 5 ? std::atoi(argv[5]) : 5;
    try {
        nz_ctx *ctx = nullptr;
        check(nz_ctx_create(sharded_rank ? std::atoi(argv[4]) : 0, &ctx), "nz_ctx_create");
        if (sharded || sharded_rank) {
            // BASELINE config 5's shape: the stock stage list on a res x res grid cut into row stripes, ghost rows over RCCL
            const int rank = sharded_rank ? std::atoi(argv[4]) : 0, world = sharded_rank ? std::atoi(argv[5]) : 1;
            const int stripes = sharded ? std::atoi(argv[4]) : world;
            const int mode = sharded ? (argc > 5 ? std::atoi(argv[5]) : NZ_HALO_EXCHANGE) : (argc > 7 ? std::atoi(argv[7]) : NZ_HALO_EXCHANGE);
            const int overlap = sharded && argc > 6 ? std::atoi(argv[6]) : 0;
            std::array<uint8_t, NZ_COMM_ID_BYTES> id{};
            if (rank == 0) {
                id = Comm::UniqueId();
                if (sharded_rank) {  // out of band: a file the other ranks wait for
                    const std::string tmp = std::string(argv[6]) + ".tmp";
                    FILE *f = std::fopen(tmp.c_str(), "wb");
                    if (!f || std::fwrite(id.data(), 1, id.size(), f) != id.size()) throw std::runtime_error("cannot write the id file");
                    std::fclose(f);
                    if (std::rename(tmp.c_str(), argv[6]) != 0) throw std::runtime_error("cannot publish the id file");
                }
            } else {
                FILE *f = nullptr;
                for (int tries = 0; tries < 1200 && !(f = std::fopen(argv[6], "rb")); tries++)
                    std::this_thread::sleep_for(std::chrono::milliseconds(50));
                if (!f || std::fread(id.data(), 1, id.size(), f) != id.size()) throw std::runtime_error("no id file from rank 0");
                std::fclose(f);
            }
            {
                Comm comm(ctx, id, rank, world);
                NoiseStage noise(ctx);
                noise.noiseType = FractalNoise::Simplex;
                noise.hurst = 0.4f;
                noise.octaves = 13;
                noise.noiseSize = 1700;
                KernelFilterStage gauss(ctx);
                gauss.filter = NZ_GAUSS5_S1;
                gauss.iterations = 17;
                FlowMapStage flow(ctx);
                flow.normMin = 0.0f;
                flow.normMax = 0.005f;
                ErosionStage erosion(ctx);
                erosion.iterations = 5;
                ShardedPipeline grid(ctx, &comm, {&noise, &gauss, &flow, &erosion}, res, res, stripes, mode, overlap, 100, 900);
                grid.Schedule().Complete();
                const std::string path = sharded_rank ? std::string(argv[2]) + "." + std::to_string(rank) : std::string(argv[2]);
                FILE *f = std::fopen(path.c_str(), "wb");
                if (!f) throw std::runtime_error("cannot open output");
                int expect = -1;
                for (int i = 0; i < grid.LocalStripes(); i++) {
                    int g0 = 0;
                    const int rows = grid.OwnedRows(i, &g0);
                    if (expect >= 0 && g0 != expect) throw std::runtime_error("local stripes are not consecutive");
                    expect = g0 + rows;
                    std::vector<float> host((size_t)rows * res);
                    grid.CopyOwnedRows(i, host.data());
                    if (std::fwrite(host.data(), sizeof(float), host.size(), f) != host.size()) throw std::runtime_error("write failed");
                }
                std::fclose(f);
                noise.OnDestroy(); gauss.OnDestroy(); flow.OnDestroy(); erosion.OnDestroy();
            }
        } else if (live) {  // BASELINE config 4's shape: cellular fBm 13 octaves -> LiveErosion.TriggerQueuedBeyerMT
            DeviceTile height(ctx, (size_t)res * res);
            NoiseStage noise(ctx);
            noise.noiseType = FractalNoise::Cellular;
            noise.hurst = 0.4f;
            noise.octaves = 13;
            noise.noiseSize = 1700;
            GeneratorData gd;
            gd.uuid = "live";
            gd.data = &height;
            gd.resolution = res;
            PipelineWorkItem wi{&gd, nullptr, nullptr, JobHandle(), nullptr};
            noise.Schedule(wi, JobHandle());
            ErosionSettings es;
            es.PARTICLES_PER_CYCLE = std::atoi(argv[4]);
            es.CYCLES = std::atoi(argv[5]);
            es.WATER_STEPS = 5;
            nz_tile_set_meta tm{};
            tm.TILE_RES[0] = tm.TILE_RES[1] = res - 16;
            tm.TILE_SIZE[0] = tm.TILE_SIZE[1] = 2000;
            tm.GENERATOR_RES[0] = tm.GENERATOR_RES[1] = res;
            tm.PATCH_RES[0] = tm.PATCH_RES[1] = 2000.0f / (float)(res - 16);
            tm.HEIGHT = 1000;
            tm.HEIGHT_F = 1000.0f;
            tm.MARGIN = 8;
            check(nz_ctx_synchronize(ctx), "nz_ctx_synchronize");
            LiveErosion erosion(ctx, &height, tm, es);
            std::vector<int> seeds;
            for (int c = 0; c < es.CYCLES; c++) seeds.push_back(11 * c + 3);
            erosion.TriggerQueuedBeyerMT(seeds).Complete();
            std::vector<float> host(3 * (size_t)res * res);
            height.CopyTo(host.data());
            erosion.poolMap.CopyTo(host.data() + (size_t)res * res);
            erosion.streamMap.CopyTo(host.data() + 2 * (size_t)res * res);
            FILE *f = std::fopen(argv[2], "wb");
            if (!f || std::fwrite(host.data(), sizeof(float), host.size(), f) != host.size()) throw std::runtime_error("write failed");
            std::fclose(f);
        } else if (context) {
            PipelineStateManager mgr(ctx);
            DeviceTile src(ctx, (size_t)res * res), dst(ctx, (size_t)res * res);
            NoiseStage noise(ctx);
            noise.noiseType = FractalNoise::Simplex;
            noise.hurst = 0.4f;
            noise.octaves = 8;
            noise.noiseSize = 200;
            WriteGeneratorContextStage wr(ctx);
            wr.contextAlias = "height";
            ReadGeneratorContextStage rd(ctx);
            rd.contextAlias = "height";
            KernelFilterStage gauss(ctx);
            gauss.filter = NZ_GAUSS5_S1;
            gauss.iterations = 3;
            BasePipeline producer({&noise, &wr}), consumer({&rd, &gauss});
            producer.contextManager = consumer.contextManager = &mgr;
            GeneratorData in, out;
            in.uuid = "src";
            in.data = &src;
            out.uuid = "out";
            out.data = &dst;
            in.resolution = out.resolution = res;
            in.xpos = out.xpos = 64;
            in.zpos = out.zpos = 32;
            consumer.Enqueue(&out);
            consumer.RunToCompletion();  // nothing to read yet: the item is parked
            if (consumer.Parked() != 1) throw std::runtime_error("consumer should wait for the context buffer");
            producer.Enqueue(&in);
            producer.RunToCompletion();
            consumer.RunToCompletion();
            if (consumer.Parked() != 0) throw std::runtime_error("consumer did not run");
            std::vector<float> host((size_t)res * res);
            dst.CopyTo(host.data());
            FILE *f = std::fopen(argv[2], "wb");
            if (!f) throw std::runtime_error("cannot open output");
            std::fwrite(host.data(), sizeof(float), host.size(), f);
            std::fclose(f);
            producer.Destroy();
            consumer.Destroy();
            mgr.OnDestroy();
        } else if (batch > 0) {
            const size_t n = (size_t)res * res;
            DeviceTile tiles(ctx, n * batch);
            std::vector<int32_t> pos(2 * batch);
            for (int k = 0; k < batch; k++) pos[2 * k] = k * res, pos[2 * k + 1] = -3 * k;
            DeviceTile dpos(ctx, (2 * batch * sizeof(int32_t) + 3) / 4);
            dpos.CopyFrom(reinterpret_cast<const float *>(pos.data()));
            NoiseStage noise(ctx);
            noise.noiseType = FractalNoise::Simplex;
            noise.hurst = 0.4f;
            noise.octaves = 13;
            noise.noiseSize = 1700;
            KernelFilterStage gauss(ctx);
            gauss.filter = NZ_GAUSS5_S1;
            gauss.iterations = 17;
            FlowMapStage flow(ctx);
            flow.normMin = 0.0f;
            flow.normMax = 0.005f;
            ErosionStage erosion(ctx);
            erosion.iterations = 5;
            BasePipeline pipe({&noise, &gauss, &flow, &erosion});
            GeneratorDataBatch gd;
            gd.uuid = "host-demo-batch";
            gd.data = &tiles;
            gd.resolution = res;
            gd.count = batch;
            gd.positions = reinterpret_cast<const int32_t *>(dpos.ptr);
            pipe.Enqueue(&gd);
            pipe.RunToCompletion();
            std::vector<float> host(n * batch);
            tiles.CopyTo(host.data());
            FILE *f = std::fopen(argv[2], "wb");
            if (!f) throw std::runtime_error("cannot open output");
            std::fwrite(host.data(), sizeof(float), host.size(), f);
            std::fclose(f);
            pipe.Destroy();
        } else if (tiles) {
            // What a COMPILED host gets out of small tiles (tools/bench_tiles.py is the same loop in Python, where the
            // interpreter's ~30 us per tile is the limit from two pipelines on): S pipelines, each on a context (stream) of
            // its own with a READ / WRITE pair, take the tiles of a row of the world in turn; nobody waits until the end.
            const int S = std::atoi(argv[4]), T = std::atoi(argv[5]);
            struct lane {
                nz_ctx *c = nullptr;
                std::unique_ptr<DeviceTile> a, b;
                std::unique_ptr<NoiseStage> noise;
                std::unique_ptr<KernelFilterStage> gauss;
                std::unique_ptr<FlowMapStage> flow;
                std::unique_ptr<ErosionStage> erosion;
                std::unique_ptr<FreeRunningPipeline> pipe;
                GeneratorData gd;
            };
            std::vector<lane> lanes(S);
            for (lane &l : lanes) {
                check(nz_ctx_create(0, &l.c), "nz_ctx_create");
                l.a.reset(new DeviceTile(l.c, (size_t)res * res));
                l.b.reset(new DeviceTile(l.c, (size_t)res * res));
                l.noise.reset(new NoiseStage(l.c));
                l.noise->noiseType = FractalNoise::Simplex;
                l.noise->hurst = 0.4f;
                l.noise->octaves = 13;
                l.noise->noiseSize = 1700;
                l.gauss.reset(new KernelFilterStage(l.c));
                l.gauss->filter = NZ_GAUSS5_S1;
                l.gauss->iterations = 17;
                l.flow.reset(new FlowMapStage(l.c));
                l.flow->iterations = 5;
                l.flow->normMin = 0.0f;
                l.flow->normMax = 0.005f;
                l.erosion.reset(new ErosionStage(l.c));
                l.erosion->iterations = 5;
                l.pipe.reset(new FreeRunningPipeline({l.noise.get(), l.gauss.get(), l.flow.get(), l.erosion.get()}));
                l.gd.uuid = "tile";
                l.gd.data = l.a.get();
                l.gd.write = l.b.get();
                l.gd.resolution = res;
            }
            auto pass = [&](int n, int x0) {
                for (int k = 0; k < n; k++) {
                    lane &l = lanes[k % S];
                    l.gd.xpos = x0 + res * k;  // a different tile of the world each time
                    l.pipe->Schedule(PipelineWorkItem{&l.gd, nullptr, nullptr, JobHandle(), nullptr});
                    l.pipe->Release();
                }
            };
            pass(3 * S, 7);
            for (lane &l : lanes) check(nz_ctx_synchronize(l.c), "nz_ctx_synchronize");
            // NZ_HOST_THREADS=1: every pipeline is driven by a host thread of its own (a context is used by one thread at a
            // time; different contexts are independent) -- HIP's ~2.5 us per command is then paid in parallel
            const bool threaded = std::getenv("NZ_HOST_THREADS") && std::atoi(std::getenv("NZ_HOST_THREADS")) > 0;
            const auto t0 = std::chrono::steady_clock::now();
            if (threaded) {
                std::vector<std::thread> th;
                std::vector<std::string> errors(S);
                for (int i = 0; i < S; i++)
                    th.emplace_back([&, i] {
                        try {
                            lane &l = lanes[i];
                            for (int k = i; k < T; k += S) {
                                l.gd.xpos = res * k;
                                l.pipe->Schedule(PipelineWorkItem{&l.gd, nullptr, nullptr, JobHandle(), nullptr});
                                l.pipe->Release();
                            }
                        } catch (const std::exception &e) {
                            errors[i] = e.what();
                        }
                    });
                for (auto &t : th) t.join();
                for (const std::string &e : errors)
                    if (!e.empty()) throw std::runtime_error(e);
            } else {
                pass(T, 0);
            }
            const auto t1 = std::chrono::steady_clock::now();
            for (lane &l : lanes) check(nz_ctx_synchronize(l.c), "nz_ctx_synchronize");
            const auto t2 = std::chrono::steady_clock::now();
            const double dt = std::chrono::duration<double>(t2 - t0).count(), host = std::chrono::duration<double>(t1 - t0).count();
            std::printf("res %5d  pipelines %2d%s: %9.1f tiles/s  %9.0f Mcells/s  (%.3f ms per tile, host enqueue %.4f ms per tile)\n", res, S,
                        threaded ? " (a host thread each)" : "", T / dt, (double)T * res * res / dt / 1e6, dt / T * 1e3, host / T * 1e3);
            // the last tile of pipeline 0, for the caller to check
            std::vector<float> host_plane((size_t)res * res);
            lanes[0].gd.data->CopyTo(host_plane.data());
            FILE *f = std::fopen(argv[2], "wb");
            if (!f) throw std::runtime_error("cannot open output");
            std::fwrite(host_plane.data(), sizeof(float), host_plane.size(), f);
            std::fclose(f);
            for (lane &l : lanes) {
                l.pipe->Destroy();
                l.a.reset();
                l.b.reset();
                nz_ctx_destroy(l.c);
            }
        } else if (reduce) {
            DeviceTile tile(ctx, (size_t)res * res);
            NoiseStage nl(ctx), nr(ctx);
            nl.noiseType = FractalNoise::Simplex;
            nl.hurst = 0.4f;
            nl.octaves = 6;
            nl.noiseSize = 300;
            nr.noiseType = FractalNoise::Cellular;
            nr.hurst = 0.5f;
            nr.octaves = 3;
            nr.noiseSize = 90;
            ReduceStage mul(ctx);
            mul.operation = 1;  // ReductionType.MULTIPLY
            BasePipeline left({&nl}), right({&nr});
            ReducePipeline red(ctx, {&mul}, &left, &right);
            GeneratorData gd;
            gd.uuid = "host-demo-reduce";
            gd.data = &tile;
            gd.resolution = res;
            gd.xpos = 37;
            gd.zpos = 11;
            int completed = 0;
            red.Enqueue(&gd, nullptr, [&](StageIO *) { completed++; });
            red.RunToCompletion();
            if (completed != 1) throw std::runtime_error("completeAction did not fire");
            std::vector<float> host((size_t)res * res);
            tile.CopyTo(host.data());
            FILE *f = std::fopen(argv[2], "wb");
            if (!f) throw std::runtime_error("cannot open output");
            std::fwrite(host.data(), sizeof(float), host.size(), f);
            std::fclose(f);
            red.Destroy();
            left.Destroy();
            right.Destroy();
        } else {
            DeviceTile tile(ctx, (size_t)res * res);
            NoiseStage noise(ctx);
            noise.noiseType = FractalNoise::Simplex;
            noise.hurst = 0.4f;
            noise.octaves = 13;
            noise.noiseSize = 1700;
            KernelFilterStage gauss(ctx);
            gauss.filter = NZ_GAUSS5_S1;
            gauss.iterations = G;
            FlowMapStage flow(ctx);
            flow.iterations = F;
            flow.normMin = 0.0f;
            flow.normMax = 0.005f;
            ErosionStage erosion(ctx);
            erosion.iterations = E;
            BasePipeline pipe({&noise, &gauss, &flow, &erosion});
            GeneratorData gd;
            gd.uuid = "host-demo";
            gd.data = &tile;
            gd.resolution = res;
            std::unique_ptr<DeviceTile> wtile;
            if (rw) {
                wtile.reset(new DeviceTile(ctx, (size_t)res * res));
                gd.write = wtile.get();
            }
            int completed = 0;
            pipe.Enqueue(&gd, nullptr, [&](StageIO *) { completed++; });
            pipe.RunToCompletion();
            if (completed != 1) throw std::runtime_error("completeAction did not fire");
            std::vector<float> host((size_t)res * res);
            gd.data->CopyTo(host.data());  // with a pair, `data` is whichever plane the last stage left the result in
            FILE *f = std::fopen(argv[2], "wb");
            if (!f) throw std::runtime_error("cannot open output");
            std::fwrite(host.data(), sizeof(float), host.size(), f);
            std::fclose(f);
            pipe.Destroy();
        }
        nz_ctx_destroy(ctx);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "host_demo: %s\n", e.what());
        return 1;
    }
    return 0;
}
