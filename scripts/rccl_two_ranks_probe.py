"""GPU box probe (one device): can the nccl (= RCCL) backend of torch.distributed be brought up with TWO ranks that share device 0?
Spawns two processes under a 127.0.0.1 rendezvous; each initialises the process group on cuda:0, all_gathers 64 doubles and runs
ncclCommInitRank through the library's own exchange (dcreg_comm_*).  Prints what worked and the first error otherwise."""
import os, sys, subprocess, socket, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "rank":
    sys.path.insert(0, ROOT)
    import torch, torch.distributed as dist
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        t = torch.full((64,), float(rank), dtype=torch.float64, device="cuda")
        out = [torch.zeros_like(t) for _ in range(2)]
        dist.all_gather(out, t)
        torch.cuda.synchronize()
        print("rank %d: torch nccl all_gather ok: %s" % (rank, [float(o[0]) for o in out]), flush=True)
    except Exception as e:
        print("rank %d: torch nccl FAILED: %s" % (rank, str(e).split("\n")[0][:300]), flush=True)
        sys.exit(3)
    sys.exit(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
procs = []
for r in range(2):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "rank"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
t0 = time.time()
for r, p in enumerate(procs):
    try:
        out, _ = p.communicate(timeout=max(5.0, 120.0 - (time.time() - t0)))
    except subprocess.TimeoutExpired:
        p.kill(); out, _ = p.communicate(); out += "\n(timed out, killed)"
    print("---- rank %d rc %s\n%s" % (r, p.returncode, "\n".join(out.strip().split("\n")[-12:])))
