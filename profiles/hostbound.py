import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
pt = ge.load_package()
scene = pt.Scene(os.path.join(os.path.dirname(ge.__file__), "scenes", "cornell.txt"))
scene.set_resolution(1280, 720)
acc = torch.zeros(1280*720*3, device="cuda")
for depth_pipe in (1, 2, 3, 4):
    pt.pathtraceFree()
    pt.pathtraceInit(scene, stream=torch.cuda.current_stream().cuda_stream, accum_dev=acc.data_ptr(), pipeline_depth=depth_pipe)
    for it in range(1, 9): pt.pathtrace(None, 0, it, readback=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 128
    for it in range(9, 9+N): pt.pathtrace(None, 0, it, readback=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("pipeline", depth_pipe, "enqueue us/step %.1f" % ((t1-t0)/N*1e6), "total us/step %.1f" % ((t2-t0)/N*1e6))
pt.pathtraceFree()
