import torch, math, sys
sys.path.insert(0,'.')
from easykv_amd import KVBank, StepPlan
torch.manual_seed(0)
for (L,Hq,H,D,T) in [(1,1,1,32,5),(1,1,1,128,5),(1,2,2,128,300),(2,4,4,32,17),(1,8,2,64,200)]:
    k = torch.randn(L,H,T,D).half().cuda(); v = torch.randn(L,H,T,D).half().cuda(); q = torch.randn(L,Hq,1,D).half().cuda()
    bank = KVBank(L,Hq,H,D,cap=T+3)
    bank.load_rows(k[:,:,:T-1], v[:,:,:T-1])
    ko, vo = bank.ordered_kv()
    print('roundtrip', (ko-k[:,:,:T-1]).abs().max().item())
    plan = StepPlan(policy='full', evict=False)
    out,_ = bank.attend(plan, q, k[:,:,T-1:].contiguous(), v[:,:,T-1:].contiguous())
    rep = Hq//H
    kk = k.float().repeat_interleave(rep,1); vv = v.float().repeat_interleave(rep,1)
    w = torch.softmax(q.float()@kk.transpose(2,3)/math.sqrt(D), -1)
    ref = w@vv
    print((L,Hq,H,D,T), 'max err', (out.float()-ref).abs().max().item())
    ko, vo = bank.ordered_kv()
    print('append ok', (ko-k).abs().max().item(), (vo-v).abs().max().item())
