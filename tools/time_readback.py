"""Host-side cost of fetching the finished 1080p image (cap_readback: mean + untile kernels + device-to-host copy into pageable
memory): the PCIe-inclusive figure quoted in DESIGN.md section 5.  Run from the repository root through gpurun."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from capsaicin_amd import capi
r = capi.Renderer(0)
r.upload_geometry(capi.Geometry("assets/cornell_box.obj")); r.upload_bluenoise(capi.load_bluenoise()); r.build_bvh()
r.set_resolution(1920, 1080); r.set_camera(capi.cornell_camera(1920, 1080))
r.render(0, 4, 8); r.sync()
for k in range(3):
    t0 = time.perf_counter(); a = r.readback(capi.BUF_ACCUM_MEAN); dt = time.perf_counter() - t0
    print("readback ACCUM_MEAN 1920x1080 float4: %.2f ms (%.1f GB/s)" % (dt * 1e3, a.nbytes / dt / 1e9))
