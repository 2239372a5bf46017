import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, os
import oracle_lib as O, hip_lib as H
from fishbirdeyevisualslam_amd import synth, ba_problem
p=synth.make_ba_problem(4000, n_kf=6, n_mp=500, n_mpb=120)
wo=int(sys.argv[1]) if len(sys.argv)>1 else 1
a,out_o,k=ba_problem.local_ba_args(p,with_odom=wo); O.call('orc_local_ba',a)
a2,out_h,k2=ba_problem.local_ba_args(p,with_odom=wo); H.call('fb_local_ba',a2)
print(np.abs(out_h['kf_Tcw']-out_o['kf_Tcw']).max())
