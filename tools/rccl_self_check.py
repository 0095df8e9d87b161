import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import __graft_entry__ as ge
sb = ge.load_package()
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
buf = sb.scenes.lattice_buffers(64, 64, jitter=1.0)
eng = sb.Engine(bounds_size=8000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=0)
eng.write_buffers(buf)
ext = torch.cuda.ExternalStream(eng.stream(), device=torch.device("cuda", 0))
eng.halo_configure([0, 1, 2], [5, 6, 7], [0], [3])
send = torch.zeros(3 * 6 + 2, device="cuda"); recv = torch.zeros(3 * 6 + 2, device="cuda")
eng.step(4)
eng.halo_pack(send.data_ptr())
with torch.cuda.stream(ext):  # same form as halo.TorchTransport (stream-ordered)
    ops = [dist.P2POp(dist.isend, send, 0), dist.P2POp(dist.irecv, recv, 0)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
eng.halo_unpack(recv.data_ptr())
eng.sync(); torch.cuda.synchronize()
out = eng.load_buffers(buf.copy())
print("self send/recv ok:", torch.equal(send, recv), "ghost0 == send5:", (out.particles[0] == out.particles[5]).all())
dist.destroy_process_group()
