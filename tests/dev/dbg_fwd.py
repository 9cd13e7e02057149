import sys; sys.path.insert(0,'.')
import torch
from tests.golden_util import load_golden, split_params
from tests.moe_gpu_util import MoeRun
from oracle.algebra_ref import AlgebraRef
meta,cfg,t = load_golden('ave_e1p1_train')
P,B = split_params(t)
run = MoeRun(cfg,P,B,t['X'],t['Y'],bf16=False,training=True).forward()
S,N,C = 6,cfg.Nx,cfg.Cx
sx = run.buf('sx', shape=(2,S,N))
print('sx err', (sx[0]-t['X'].sum(-1)).abs().max(), (sx[1]-(t['X']**2).sum(-1)).abs().max())
A = AlgebraRef(cfg,P,B); A.forward(t['X'],t['Y'],True)
rmu = run.buf('rmu', shape=(2,S,N,cfg.E))
print('r gpu', rmu[0,0,:4], 'ref', A.sv['E'][0]['r'][0,:4], A.sv['E'][1]['r'][0,:4])
print('mu gpu', rmu[1,0,:4], 'ref', A.sv['E'][0]['mu'][0,:4], A.sv['E'][1]['mu'][0,:4])
ws = run.buf('wsum'); print('wsum', ws[:32])
print('ref wsum e0', A.sv['E'][0]['wsum'], 'e1', A.sv['E'][1]['wsum'])
Ts = run.buf('Tsum', shape=(2,S,cfg.K+2)); print('Tsum', Ts[0,0], 'ref tbar*C', A.sv['E'][0]['tbar'][0]*C)
