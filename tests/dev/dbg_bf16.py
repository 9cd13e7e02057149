import sys; sys.path.insert(0,'.')
import torch
from tests.golden_util import load_golden, split_params
from tests.moe_gpu_util import MoeRun
for name in ['ave_train','ave_wide_train','avs_v2_train']:
    meta,cfg,t = load_golden(name)
    P,B = split_params(t)
    run = MoeRun(cfg,P,B,t['X'],t['Y'],bf16=True,training=True,noise=t.get('noise')).forward()
    out = run.out.float().cpu()
    print(name, 'out normrel', float((out-t['out']).norm()/t['out'].norm()), 'maxrel', float((out-t['out']).abs().max()/t['out'].abs().max()))
    g = run.backward(t['grad_out'], lb_weight=meta['lb_weight'])
    rows=[]
    for k,v in g.items():
        ref=t['grad.'+k]; n=float(ref.norm())
        if n>0: rows.append((float((v-ref).norm())/n, k, float((v-ref).abs().max()/ref.abs().max())))
    rows.sort(reverse=True)
    for r in rows[:8]: print('   %-45s normrel %.3e maxrel %.3e'%(r[1],r[0],r[2]))
