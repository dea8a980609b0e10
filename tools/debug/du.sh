python tests/dropin_user.py a 2>/dev/null | grep DROPIN_USER | python -c "
import sys, json, numpy as np
d = json.loads(sys.stdin.read().split('DROPIN_USER ',1)[1])
g = np.load('tests/golden/g19_trainer_loop.npz')
for it, s in enumerate(d['steps']):
    for k in ('loss_ce','loss_dice','unsup_loss','reco_loss','loss_eqv','loss_q','loss'):
        print(it, k, s[k], float(g[f'a_{it}_{k}']), abs(s[k]-float(g[f'a_{it}_{k}'])))
    print(it, 'probe', s['probe'], g[f'a_{it}_probe'].tolist(), 'bank', s['bank_len'], g[f'a_{it}_bank_len'].tolist())
for k,v in d['end'].items(): print(k, v, float(g['a_end_'+k]), abs(v/float(g['a_end_'+k])-1))
"
