import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import torch
from util import p2_opts, rel_err
from cl4wsis_amd import segmentation_module as sm
from oracle import weights, model as om
gpu=torch.device('cuda:0'); crop=65
opts=p2_opts(crop); model=sm.make_model(opts, classes=[16,5]); sd=weights.fill_(model.state_dict(), seed=1); model.load_state_dict(sd); model.to(gpu)
model.train(); model.body.eval(); model.head.eval(); model.cls.eval()
g=torch.Generator().manual_seed(9); x=torch.randn(3,3,crop,crop,generator=g)
def run(dt):
    osd={k:(v.to(dt) if v.is_floating_point() else v.clone()) for k,v in sd.items()}
    names=[k for k in osd if (k.startswith("decoder.") or k.startswith("instance_head.")) and osd[k].is_floating_point() and not k.endswith(("running_mean","running_var"))]
    for k in names: osd[k].requires_grad_(True)
    with torch.no_grad(): _,aux=om.forward_seg(x.to(dt),osd,pooling=opts.pooling)
    feats={k:v.detach() for k,v in aux["features"].items()}
    gg=torch.Generator().manual_seed(10)
    keep=(torch.rand(3,256,feats["res5"].shape[2],feats["res5"].shape[3],generator=gg)>=0.5)
    ins=om.instance_decoder(feats,osd,training=True,dropout_mask=keep.to(dt)); ref=om.instance_head(ins,osd,training=True)
    gc=torch.randn(ref["center"].shape,generator=gg); go=torch.randn(ref["offset"].shape,generator=gg)
    ((ref["center"]*gc.to(dt)).sum()+(ref["offset"]*go.to(dt)).sum()).backward()
    return names,{k:osd[k].grad.double() for k in names},keep,gc,go
names,g64,keep,gc,go=run(torch.float64)
_,g32,_,_,_=run(torch.float32)
_,f=model.forward_seg(x.to(gpu),interpolate=False); br=model.instance_branch()
pred,tape=br.forward_train(f["features"],dropout_keep=keep.to(torch.uint8).to(gpu))
grads=br.backward(tape,{"center":gc.to(gpu),"offset":go.to(gpu)}); torch.cuda.synchronize()
params=dict(model.named_parameters())
for k in names:
    eh=rel_err(grads[id(params[k])].cpu().double(),g64[k]); ec=rel_err(g32[k],g64[k])
    if eh>2e-4 or ec>2e-4: print(f"{k:60s} hip {eh:.2e} cpu32 {ec:.2e}")
