import sys, time, os
sys.path.insert(0,'.')
from oracle.oracle import Oracle
from godot_atmosphere_shader_amd.demo import *
from godot_atmosphere_shader_amd import scene as S
print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
for f in ('/sys/fs/cgroup/cpu.max','/sys/fs/cgroup/cpu/cpu.cfs_quota_us','/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
o=Oracle('f32_fast')
cam=S.Camera.from_pose(1920,1080,'P_space'); depth=S.depth_ground_sphere(cam)
tex=demo_textures(); params=demo_params()
for nt in (1,8,16,32,64,128,256):
    t=time.time(); o.render(params, dict(tex, optical_depth=None), CONFIGS['no_clouds_32x8_direct'][1], demo_frame(cam), depth, nthreads=nt); dt=time.time()-t
    print(nt, 'threads %.2f Mrays/s' % (1920*1080/dt/1e6))
