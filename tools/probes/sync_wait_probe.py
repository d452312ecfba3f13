import ctypes, os, sys, time
mode = sys.argv[1]
hip = ctypes.CDLL("libamdhip64.so")
if mode == "spin":
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(1))
elif mode == "yield":
    print("hipSetDeviceFlags(yield) ->", hip.hipSetDeviceFlags(2))
elif mode == "block":
    print("hipSetDeviceFlags(blocking) ->", hip.hipSetDeviceFlags(4))
import torch
sys.path.insert(0, "/root/repo")
from godot_atmosphere_shader_amd import scene as S
from godot_atmosphere_shader_amd.demo import demo_textures, make_node
tex = demo_textures()
cam = S.Camera.from_pose(1920, 1080, "P_space")
depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
node = make_node("no_clouds_32x8_direct", tex)
out = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda")
frame = node.prepare_frame(cam)
s = torch.cuda.current_stream().cuda_stream
for _ in range(300):
    node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), s)
torch.cuda.synchronize()
res = []
for rep in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 20 * 1e3)
res.sort()
# an empty sync's cost
e = []
for _ in range(50):
    t0 = time.perf_counter(); torch.cuda.synchronize(); e.append((time.perf_counter() - t0) * 1e6)
e.sort()
print(f"{mode}: 20-step regions, ms per step: median {res[len(res)//2]:.4f}  min {res[0]:.4f}  max {res[-1]:.4f};  an idle torch.cuda.synchronize(): median {e[len(e)//2]:.1f} us")
