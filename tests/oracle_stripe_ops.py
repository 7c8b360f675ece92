"""Stripe operations of noize_job_amd.sharded implemented on the CPU oracle (test infrastructure).

Lets the sharding schedule (partition, halo widths, exchange order) be exercised without a GPU:
the multi-process `gloo` test drives exactly the host logic the GPU path runs, with the oracle as
the compute back end.  Buffers are torch CPU float32 tensors."""
import numpy as np

import oracle as O

FUSE_CAP = 3  # same default fusion depth as the library uses for 5-tap kernels


class OracleStripeOps:
    def kernel_filter_halo_rows(self, filter, iterations):
        return iterations * ((O.kernel_filter_table(filter)[3] - 1) // 2)

    def kernel_filter_max_fused(self, filter):
        return FUSE_CAP

    def erosion_max_fused(self):
        return 16

    @staticmethod
    def _valid(plan):
        """rows of the buffer that lie inside the global grid: the oracle clamps at its plane edge, so
        the plane handed to it must end exactly at the global border"""
        return max(0, -plan.grow0), min(plan.rows, plan.grows - plan.grow0)

    def fractal(self, buf, plan, p):
        a = O.fractal(p.noiseType, plan.nown, plan.cols, p.hurst, p.startingAmplitude, p.stepdown, p.detuneRate,
                      p.octaves, p.xpos, p.zpos + plan.g0, p.noiseSize)
        buf.numpy()[plan.own0:plan.own1] = a

    def kernel_filter(self, src, dst, plan, filter, T):
        v0, v1 = self._valid(plan)
        out = O.kernel_filter(src.numpy()[v0:v1], filter, T)
        dst.numpy()[plan.own0:plan.own1] = out[plan.own0 - v0:plan.own1 - v0]

    def erosion(self, src, dst, plan, E):
        v0, v1 = self._valid(plan)
        out = O.erosion_min(src.numpy()[v0:v1], E)
        dst.numpy()[plan.own0:plan.own1] = out[plan.own0 - v0:plan.own1 - v0]

    def flow_fused_max(self):
        return 3

    def flow_fused(self, h, S_in, S_out, dst, plan, n, first, last, normMin, normMax):
        v0, v1 = self._valid(plan)
        hv = h.numpy()[v0:v1]
        if first:
            w = np.full_like(hv, 0.0001)
            fl = [np.zeros_like(hv) for _ in range(4)]
        else:
            w = S_in[0].numpy()[v0:v1]
            fl = [S_in[i].numpy()[v0:v1] for i in range(1, 5)]
        for it in range(n):
            fl = O.flow_step(hv, w, *fl)
            if not (last and it == n - 1):
                w = O.water_step(w, *fl)
        sl = slice(plan.own0 - v0, plan.own1 - v0)
        if last:
            out = O.normalize(O.velocity(*fl), normMin, normMax)
            dst.numpy()[plan.own0:plan.own1] = out[sl]
        else:
            S_out[0].numpy()[plan.own0:plan.own1] = w[sl]
            for i in range(4):
                S_out[1 + i].numpy()[plan.own0:plan.own1] = fl[i][sl]

    def map_range(self, buf, n, res, lim_min=float("inf"), lim_max=float("-inf")):
        res.numpy()[:] = O.get_map_range(buf.numpy().reshape(-1)[:n], lim_min, lim_max)

    def normalize_args(self, buf, plan, args):
        own = buf.numpy()[plan.own0:plan.own1]
        own[:] = O.normalize_args(own, args.numpy())
