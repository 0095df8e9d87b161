import time, sys
sys.path.insert(0,'.')
import __graft_entry__ as ge
sb=ge.load_package()
buf=sb.scenes.lattice_buffers(1000,1000,d=30.0,origin=(1000.0,1000.0),jitter=1.0,layout=2)
for mode,blk in ((0,0),(2,0)):
    eng=sb.Engine(bounds_size=32000.0,layout=2,max_particles=buf.max_particles,max_beams=buf.max_beams,collision_mode=mode,block_substeps=blk)
    for k in range(2):
        t=time.perf_counter(); eng.write_buffers(buf); print("mode",mode,"write_buffers %.1f ms"%((time.perf_counter()-t)*1e3),flush=True)
    eng.step(8); eng.sync()
    out=buf.copy()   # destination allocated outside the timer
    for k in range(2):
        t=time.perf_counter(); eng.load_buffers(out); print("load_buffers %.1f ms"%((time.perf_counter()-t)*1e3),flush=True)
    eng.destroy()
