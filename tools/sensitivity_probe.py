import os, sys, torch
sys.path.insert(0, '/root/repo')
from dvd_gan_amd.gen_net import Generator
def rel(a,b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
dev='cuda'; B,T,ch,ncls=4,48,32,101
torch.manual_seed(0); z=torch.randn(B,120,device=dev); cls=torch.randint(0,ncls,(B,),device=dev)
outs={}
for name,dt,noise in (("f32",torch.float32,0),("f32+w*2^-9 noise",torch.float32,2**-9),("f32+1e-6 noise",torch.float32,1e-6),("bf16",torch.bfloat16,0)):
    torch.manual_seed(1)
    G=Generator(120,4,ncls,ch,T,compute_dtype=dt).to(dev).train()
    if noise:
        torch.manual_seed(5)
        with torch.no_grad():
            for k,p in G.named_parameters():
                if p.requires_grad: p.mul_(1+noise*torch.randn_like(p))
    with torch.no_grad(): outs[name]=G(z,cls)
    print(name, "out abs mean", float(outs[name].abs().mean()), "std", float(outs[name].std()))
for k in outs:
    print(k, "rel vs f32:", rel(outs[k], outs["f32"]))
