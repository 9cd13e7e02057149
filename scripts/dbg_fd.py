import sys; sys.path.insert(0,'.')
import torch
from tests.test_fullsize_properties_gpu import _site, _inputs, _fwd, CFG2
dev=torch.device('cuda:0')
for training in (False, True):
  for S in (4, 64):
    m=_site(dev).train(training)
    X,Y=_inputs(dev,S=S)
    gen=torch.Generator(device=dev).manual_seed(11)
    G=torch.randn(S,CFG2['N_a'],CFG2['C'],device=dev,generator=gen)
    vX=torch.randn(X.shape,device=dev,generator=gen); vY=torch.randn(Y.shape,device=dev,generator=gen)
    Xr,Yr=X.clone().requires_grad_(True),Y.clone().requires_grad_(True)
    out,_=_fwd(m,Xr,Yr); (out*G).sum().backward()
    anaX=float((Xr.grad*vX).sum()); anaY=float((Yr.grad*vY).sum())
    res=[]
    for h in (1e-2,2e-3,5e-4):
        with torch.no_grad():
            fpx=float((_fwd(m,X+h*vX,Y)[0].double()*G.double()).sum()); fmx=float((_fwd(m,X-h*vX,Y)[0].double()*G.double()).sum())
            fpy=float((_fwd(m,X,Y+h*vY)[0].double()*G.double()).sum()); fmy=float((_fwd(m,X,Y-h*vY)[0].double()*G.double()).sum())
        res.append((h,(fpx-fmx)/(2*h),(fpy-fmy)/(2*h)))
    print('training',training,'S',S,'anaX %.2f anaY %.3f'%(anaX,anaY), ['h=%g numX %.2f numY %.3f'%r for r in res])
