import sys; sys.path.insert(0, '.')
import numpy as np
from mr_gan_amd import engine as E
o = E.debug_tr_probe()
np.set_printoptions(linewidth=200)
print("linear probe (lane: 4 received element indices)")
for l in range(64): print(l, o[0][l][:4])
print("ks_frag probe (lane: row,col pairs)")
for l in range(64): print(l, [(int(v)>>8, int(v)&255) for v in o[1][l]])
