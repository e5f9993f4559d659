import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
from oracle import synth, labels as L
import test_label_gpu as T
seed=int(sys.argv[1]) if len(sys.argv)>1 else 0
gpu=torch.device('cuda:0')
inp=synth.label_stage_inputs(seed,b=2+seed%2,H=129,W=129)
ref=T._run_oracle(inp); got=T._run_hip(inp,gpu)
print(got['counters'])
ro=ref['refined']['offset']; go=got['refined']['offset'].cpu()
rw=ref['refined']['weight']; gw=got['refined']['weight'].cpu()
mo=(ro!=go).any(1); print('offset mismatch px', int(mo.sum()), 'weight>0 ref', int((rw>0).sum()), 'got', int((gw>0).sum()))
seg=ref['seg_gt']
for b in range(seg.shape[0]):
    for v in np.unique(seg[b].numpy()):
        if v==0: continue
        n,lab,area,cent=L.connected_components((seg[b].numpy()==v).astype(np.uint8),8)
        for k in range(1,n):
            sel=torch.from_numpy(lab==k)
            mm=int((mo[b]&sel).sum())
            print(f'b{b} cls{v} comp{k} area{area[k]} mism{mm} refw>0 {int(((rw[b,0]>0)&sel).sum())} gotw>0 {int(((gw[b,0]>0)&sel).sum())}')
            if mm:
                ys,xs=np.nonzero((mo[b]&sel).numpy())
                for y,x in list(zip(ys,xs))[:5]:
                    print('   px',y,x,'ref off',ro[b,:,y,x].tolist(),'got',go[b,:,y,x].tolist(),'w',rw[b,0,y,x].item(),gw[b,0,y,x].item())
print('---- per-contour instance points')
H,W=129,129
yy,xx=torch.meshgrid(torch.arange(H).float(),torch.arange(W).float(),indexing='ij')
for name,off,w in (('ref',ro,rw),('got',go,gw)):
    for b in range(seg.shape[0]):
        m=w[b,0]>0
        py=(off[b,0]+yy)[m]; px=(off[b,1]+xx)[m]; ww=w[b,0][m]
        u=torch.unique(torch.stack([py,px,ww],1),dim=0,return_counts=True)
        print(name,b,[(int(a[0]),int(a[1]),round(float(a[2]),4),int(c)) for a,c in zip(*u)])
cen=torch.nn.functional.interpolate(inp['center_lo'],size=(H,W),mode='bilinear',align_corners=False)
print('center at (124,110) cls channels', cen[0,15:,124,110].tolist())
print('center at (127,107)', cen[0,15:,127,107].tolist())
