import sys, numpy as np, torch
sys.path.insert(0, '.')
from esac_amd import api, synthetic as S
eng = api.engine(0)
tot = np.zeros(24)
n = 0
for k in range(16):
    f = S.make_frame(k); ha = S.gating_assignment(f, 256)
    p = eng.make_params(1, 60, 80, 256, call=k)
    for rep in range(3):
        eng.forward_device(torch.from_numpy(f['coords']).cuda(), torch.from_numpy(ha).cuda(), p)
    tot += eng.read(api.BUF_CYCLES)[:24]; n += 1
tot /= n
names = ['total','argmax','error_images','epilogue','rodrigues+chain','point_loop','block_sum','transform','solve','passes','ep_loads','ep_screen','ep_exact','ep_stores','ep_compact','ep_exact_trips','lm_control','rejected_trials','pinv_steps','-','-','-','-','-']
for a,b in zip(names, tot): print('%-16s %10.0f cycles  %7.2f us' % (a, b, b/2400.))
print('per pass: rod %.0f pts %.0f sum %.0f tr %.0f solve %.0f' % tuple(tot[i]/tot[9] for i in (4,5,6,7,8)))
acc = sum(tot[i] for i in (1,2,3,4,5,6,7,8,16))
print('sections 1-8,16: %.0f cycles = %.1f %% of total; unaccounted %.2f us' % (acc, 100*acc/tot[0], (tot[0]-acc)/2400.))
