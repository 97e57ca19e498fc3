import sys, time, torch
sys.path.insert(0,'.')
from easykv_amd import KVBank, StepPlan, geometry
L,Hq,H,D = 32,32,int(sys.argv[2]) if len(sys.argv)>2 else 32,128
import os
S, stride, budget = int(os.environ.get('S','4096')), int(sys.argv[1]) if len(sys.argv)>1 else 8, float(os.environ.get('BUDGET','0.5'))
STREAM = os.environ.get('STREAM','0')=='1'
NSPLIT = int(sys.argv[3]) if len(sys.argv)>3 else 0
import os
POLICY = os.environ.get('POLICY','roco'); NOEVICT = os.environ.get('NOEVICT','0')=='1'
bp, idx, r_idx = geometry("encoding", S, budget, stride)
recent, sink = int(bp*0.1), 4
dev=torch.device('cuda'); g=torch.Generator(device=dev).manual_seed(0)
bank=KVBank(L,Hq,H,D,cap=idx+stride)
if STREAM:
    from easykv_amd.api import rope_tables
    bank.set_rope(*rope_tables(idx+stride+8, D))
def rnd(h,n): return torch.randn(L,h,n,D,generator=g,device=dev).half()
# dense prefix, layer blocks of 8 to bound the workspace; inputs generated up front, first pass = warm-up on a scratch bank
qp = [tuple(rnd(h, r_idx)[:8].contiguous() for h in (Hq, H, H)) for _ in range(L // 8)]
for timed in (False, True):
    b = bank if timed else KVBank(L,Hq,H,D,cap=idx+stride)
    if STREAM and not timed:
        from easykv_amd.api import rope_tables
        b.set_rope(*rope_tables(idx+stride+8, D))
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i, l0 in enumerate(range(0,L,8)):
        q,k,v = qp[i]
        b.attend(StepPlan(policy='full',phase='prefill',accumulate=False,streaming=STREAM), q,k,v, layer_begin=l0)
    torch.cuda.synchronize(); t_prefix=time.perf_counter()-t0
    del b
bank.state_init(idx+stride,2,stride)
n_chunks=(S-r_idx)//stride
qs=[rnd(Hq,stride) for _ in range(4)]; ks=[rnd(H,stride) for _ in range(4)]; vs=[rnd(H,stride) for _ in range(4)]
ev=[]
torch.cuda.synchronize(); t0=time.perf_counter()
for c in range(n_chunks):
    t_now=bank.n_slots[0]+stride
    if NOEVICT and t_now+stride>bank.cap: break
    plan=StepPlan(streaming=STREAM,policy=POLICY,phase='prefill',accumulate=t_now>idx,evict=(t_now>idx and not NOEVICT and c<8) if NOEVICT else t_now>idx,budget=bp,recent=recent,sink=sink,stride=stride,tova_head_mean=True,n_split=NSPLIT)
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e2=torch.cuda.Event(enable_timing=True)
    e0.record(); bank.attend(plan,qs[c%4],ks[c%4],vs[c%4],phases=1); e1.record(); bank.attend(plan,qs[c%4],ks[c%4],vs[c%4],phases=2); e2.record()
    ev.append((e0,e1,e2))
torch.cuda.synchronize(); t_chunks=time.perf_counter()-t0
# whole step as the library runs it (phases=0: ONE launch when the scorer fuses into the attention kernel)
ev0=[]
for c in range(40):
    plan=StepPlan(streaming=STREAM,policy=POLICY,phase='prefill',accumulate=True,evict=not NOEVICT,budget=bp,recent=recent,sink=sink,stride=stride,tova_head_mean=True,n_split=NSPLIT)
    if NOEVICT: break
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); bank.attend(plan,qs[c%4],ks[c%4],vs[c%4]); e1.record(); ev0.append((e0,e1))
torch.cuda.synchronize()
if ev0:
    tw=sum(a.elapsed_time(b) for a,b in ev0[4:])/len(ev0[4:])*1e3
    print(f"whole step, phases=0 (n_split,fused-plan)={bank.step_plan(plan,stride)}: {tw:.1f} us -> {stride/tw*1e6:.0f} prefill tok/s")
ta=sum(a.elapsed_time(b) for a,b,_ in ev[2:])/len(ev[2:])*1e3; ts=sum(b.elapsed_time(c) for _,b,c in ev[2:])/len(ev[2:])*1e3
T=idx+stride
bytes_attn = L*(2*H*T*D*2 + 2*Hq*stride*D*2 + 2*H*stride*D*2)
print(f"stride {stride} H {H} n_split {NSPLIT}: geometry bp={bp} idx={idx} r_idx={r_idx} T={T}; dense prefix {r_idx} tok x {L} layers: {t_prefix*1e3:.1f} ms")
print(f"chunk step (32 layers/launch): attn {ta:.1f} us ({bytes_attn/ta/1e3:.0f} GB/s alg) + score/select {ts:.1f} us ; {n_chunks} chunks in {t_chunks*1e3:.1f} ms -> {stride*n_chunks/t_chunks:.0f} prefill tok/s (chunk phase)")
