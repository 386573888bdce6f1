"""Exercises the CPU-side native code (oracle, host loader / image writers) inside ONE process whose libraries were built with
-fsanitize=address,undefined: every scene, the README extras, mesh scenes, malformed scene and OBJ files, golden primitives.
Run by tests/run_sanitizers.sh (GPU sanitizers are not available on the pool; the HIP library is not part of this run)."""
import os, sys, glob, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("PT_SAN_DIR", "/tmp/pt_san")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/oracle')
import oracle as o
o.LIB_PATH=OUT + '/libptoracle.so'
o.build=lambda *a,**k: None
import __graft_entry__ as ge
pt=ge.load_package()
pt.HOST_LIB_PATH=OUT + '/libpt_host.so'
for f in sorted(glob.glob(ROOT+'/scenes/*.txt')):
    a=pt.Scene(f); b=o.Scene(f)
    assert a.geoms.tobytes()==b.geoms.tobytes(), f
    a.set_resolution(40,30); b.set_resolution(40,30)
    r=o.Renderer(b.camera,b.geoms,b.materials,5,meshes=b.meshes)
    img=np.zeros(40*30*3,np.float32)
    for it in (1,2): r.iterate(it,img)
    r.set_extras(0.2,9.0,True)
    r.iterate(3,img)
    od=r.dump_paths(1,2)
    print(os.path.basename(f), float(img.sum()), len(od[3]))
    pt.host_lib().pth_save_png(((OUT + '/x_'+os.path.basename(f)).encode()), img.ctypes.data, 40, 30, 3.0)
    pt.host_lib().pth_save_hdr(((OUT + '/x_'+os.path.basename(f)).encode()), img.ctypes.data, 40, 30, 3.0)
# broken inputs
open(OUT + '/bad.obj','w').write('v 1 2\nf 1 2 3\nf\nv\nf 0 0 0\nf -9 1 2\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/ 2// 3/1/1 1\n')
open(OUT + '/bad.txt','w').write('OBJECT 0\nmesh bad.obj\nmaterial 7\nTRANS 1\n\nOBJECT 1\nmesh\n\nMATERIAL 0\nRGB\n\nCAMERA\nRES 4\n\n\nOBJECT 5\ncube\n')
for mod in (pt,o):
    s=mod.Scene(OUT + '/bad.txt'); print(len(s.geoms), {k:v.shape for k,v in s.meshes.items()})
# golden primitives
z=np.load(ROOT+'/tests/golden/intersections.npz')
G=np.frombuffer(z['geoms'].tobytes(), o.GEOM_DTYPE)
for gi in range(len(G)):
    for i in range(0,200):
        o.intersect(G[gi:gi+1], z['rays'][gi][i])
t=np.load(ROOT+'/tests/golden/triangles.npz')
for i in range(2000): o.mesh_triangle(t['origin'][i],t['direction'][i],t['v'][i,0],t['v'][i,1],t['v'][i,2])
print(o.scan_exclusive(np.arange(1000)%3)[-1], len(o.compact_nonzero(np.arange(1000)%3)))
print('sanitizer run finished')
