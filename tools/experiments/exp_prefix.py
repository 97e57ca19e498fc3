"""Dense / scored dense prefix of the wide-block kernel alone, warmed (TFLOP/s); EASYKV_HIP_LIB selects the build."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan
dev = torch.device("cuda")
def prefix(n, L=32, H=32, D=128, reps=6):
    g = torch.Generator(device=dev).manual_seed(2)
    q, k, v = (torch.randn(L, H, n, D, generator=g, device=dev).half() for _ in range(3))
    out = torch.empty(L, H, n, D, dtype=torch.float16, device=dev)
    ms = []
    for _ in range(reps + 2):
        bank = KVBank(L, H, H, D, cap=n + 8, device=dev)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record(); bank.attend(StepPlan(policy="full", phase="prefill", accumulate=False), q, k, v, out=out); e[1].record()
        torch.cuda.synchronize(); ms.append(e[0].elapsed_time(e[1])); del bank
    t = sum(ms[2:]) / reps
    return round(t, 2), round(4.0 * H * D * n * n / 2 * L / (t * 1e-3) / 1e12, 1)
print(os.path.basename(os.environ.get("EASYKV_HIP_LIB", "default")), "prefix 4906:", prefix(4906), "2048:", prefix(2048), flush=True)
