"""Cost of one torch.distributed (RCCL) collective issued from PyTorch with a world of one: host issue time and stream time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-gs_amd")]
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import lidargs_dist
comm = lidargs_dist.TorchDistComm()
N = 64 * 2650
cases = {"all_gather T_pass (0.7 MB)": lambda: comm.all_gather(torch.empty(N, device="cuda")),
         "all_gather planes (3.4 MB)": lambda: comm.all_gather(torch.empty(5, N, device="cuda")),
         "all_reduce radii (8 MB)": lambda: comm.all_reduce(torch.zeros(2_000_000, dtype=torch.int32, device="cuda")),
         "all_to_all rows (18 MB)": lambda: comm.all_to_all_rows(torch.empty(250_000, 18, device="cuda"), [250_000], [250_000])}
x = torch.empty(1 << 20, device="cuda")
for name, fn in cases.items():
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(50):
        fn(); x.add_(1.0)                     # a dependent-free compute kernel between collectives, as in a frame
    e1.record(); t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name:32s} host issue {1e6 * t_host / 50:7.1f} us/call   stream {1e3 * e0.elapsed_time(e1) / 50:7.1f} us/call")
dist.destroy_process_group()
