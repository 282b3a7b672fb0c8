import sys; sys.path.insert(0,'/root/repo')
import torch, torch.nn.functional as F
from oracle import get_cpu as O
torch.set_num_threads(8)
L,B,N,seed=12,4,200,3
sd=O.fill_state_dict("finetune",seed=seed,num_layers=L); batch=O.make_batch(B,N,seed)
leaf={k:v.clone().requires_grad_(True) for k,v in sd.items()}
x=O.embed(batch["region_motif"],leaf)
caps={}
for i in range(L):
    p=f"encoder.blocks.{i}."
    h=F.layer_norm(x,(768,),leaf[p+"norm1.weight"],leaf[p+"norm1.bias"],1e-6)
    Bq,T,C=h.shape
    qkvb=torch.cat((leaf[p+"attn.q_bias"],torch.zeros(768),leaf[p+"attn.v_bias"]))
    qkv=F.linear(h,leaf[p+"attn.qkv.weight"],qkvb); qkv.retain_grad()
    q5=qkv.reshape(Bq,T,3,12,64).permute(2,0,3,1,4); q,k,v=q5[0],q5[1],q5[2]
    attn=((q*0.125)@k.transpose(-2,-1)).softmax(-1)
    o=(attn@v); o.retain_grad()
    a=F.linear(o.transpose(1,2).reshape(Bq,T,C),leaf[p+"attn.proj.weight"],leaf[p+"attn.proj.bias"])
    x=x+a
    h2=F.layer_norm(x,(768,),leaf[p+"norm2.weight"],leaf[p+"norm2.bias"],1e-6)
    x=x+O.mlp(h2,leaf,p)
    caps[i]=(qkv,o)
xf=F.layer_norm(x,(768,),leaf["encoder.norm.weight"],leaf["encoder.norm.bias"],1e-6)[:,1:]
pred=F.softplus(F.linear(xf,leaf["head_exp.head.weight"],leaf["head_exp.head.bias"]))
loss=O.poisson_nll(pred,batch["exp_label"]); loss.backward()
bf=lambda t:t.to(torch.bfloat16).float()
def dqb(q,k,v,do,o_for_delta, round_ds=False):
    s=(q*0.125)@k.transpose(-2,-1); p=s.softmax(-1)
    dp=do@v.transpose(-2,-1); delta=(do*o_for_delta).sum(-1,keepdim=True)
    ds=p*(dp-delta)
    if round_ds: ds=bf(ds)
    dq=(ds@k)*0.125
    return dq.sum((0,2)).reshape(-1)   # [H*64]
for i in (0,3,9,11):
    qkv,o=caps[i]; ref=leaf[f"encoder.blocks.{i}.attn.q_bias"].grad
    q5=qkv.detach().reshape(B,N+1,3,12,64).permute(2,0,3,1,4); q,k,v=q5[0],q5[1],q5[2]
    do=o.grad; oo=o.detach()
    err=lambda g:((g-ref).norm()/ref.norm()).item()
    print(i, "exact",err(dqb(q,k,v,do,oo)), "roundO_delta",err(dqb(q,k,v,do,bf(oo))), "round_dO",err(dqb(q,k,v,bf(do),oo)),
          "round_qkv",err(dqb(bf(q),bf(k),bf(v),do,oo)), "round_dS", err(dqb(q,k,v,do,oo,True)),
          "all", err(dqb(bf(q),bf(k),bf(v),bf(do),bf(oo),True)),
          "all_consistentO", err((lambda qq,kk,vv,dd: dqb(qq,kk,vv,dd,((qq*0.125)@kk.transpose(-2,-1)).softmax(-1)@vv))(bf(q),bf(k),bf(v),bf(do))))
print("---- forward-kernel-like O (P rounded to bf16 inside the PV product)")
for i in (0,3,9,11):
    qkv,o=caps[i]; ref=leaf[f"encoder.blocks.{i}.attn.q_bias"].grad
    q5=qkv.detach().reshape(B,N+1,3,12,64).permute(2,0,3,1,4); q,k,v=bf(q5[0]),bf(q5[1]),bf(q5[2])
    do=bf(o.grad)
    s=(q*0.125)@k.transpose(-2,-1); mx=s.max(-1,keepdim=True).values; pt=torch.exp(s-mx); l=pt.sum(-1,keepdim=True)
    o32=(bf(pt)@v)/l
    hi=bf(o32); lo=bf(o32-hi)
    err=lambda g:((g-ref).norm()/ref.norm()).item()
    print(i,"delta from bf16 O",err(dqb(q,k,v,do,hi,True)),"from fwd fp32 O",err(dqb(q,k,v,do,o32,True)),"hi+lo",err(dqb(q,k,v,do,hi+lo,True)))
