import sys, time
sys.path.insert(0,'.')
import torch, numpy as np
from godot_atmosphere_shader_amd.demo import *
from godot_atmosphere_shader_amd import scene as S
tex=demo_textures(); params=demo_params()
cam=S.Camera.from_pose(1920,1080,'P_space'); depth=torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
for wl in ['no_clouds_32x8_direct','no_clouds_8']:
    node=make_node(wl, tex, params)
    out=torch.empty((1080,1920,4),device='cuda')
    frame=node.prepare_frame(cam)
    ref=node.render(cam, depth).clone(); torch.cuda.synchronize()
    K=50
    # eager
    s=torch.cuda.current_stream().cuda_stream
    for _ in range(10): node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(K): node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize(); te=(time.perf_counter()-t)/K
    # graph
    g=torch.cuda.CUDAGraph()
    side=torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(K): node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t=time.perf_counter(); g.replay(); torch.cuda.synchronize(); tg=(time.perf_counter()-t)/K
    print(wl, 'eager %.4f ms/step  graph %.4f ms/step'%(te*1e3, tg*1e3), 'equal', torch.equal(out, ref))
    node.close()
