import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import noize_job_amd as nj, oracle as O
f32=np.float32
with nj.Context(0) as ctx:
    for res, it in ((600, 6), (1024, 17), (4096, 17), (4096, 9)):
        rng=np.random.default_rng(res+it)
        t=rng.random((res,res),dtype=f32)
        want=O.kernel_filter(t,2,it)
        for rep in range(2):
            d=nj.GeneratorData("g",ctx.from_host(t),res,0,0)
            st=nj.KernelFilterStage(ctx,nj.KernelFilterType.Gauss5_S1,it)
            t0=time.time()
            try:
                st.ReceiveHandledInput(nj.PipelineWorkItem(d),nj.JobHandle())
                st.jobHandle.Complete()
            except Exception as e:
                print("ERROR", e)
            got=d.data.ToArray((res,res))
            print(res,it,rep,"mismatch",int((got!=want).sum()),"%.1f ms"%((time.time()-t0)*1e3), flush=True)
            d.data.Dispose(); st.OnDestroy()
