"""Conditioning-image path at the BFM's size: 35,721 vertices / 70,688 triangles (the external
BFM_model_front.mat has 35,709 / ~70 k), 224x224 frames, clips of T frames.  Times the two C-ABI calls with HIP events on
the launch stream, reports the HBM roofline of the reconstruction (bases read once per clip) and, beside it, the CPU
checkers on the same mesh: the reference's compiled rasteriser (oracle/_ref, kind "reference") when it travelled with the
repo, and the numpy restatement of the reconstruction (kind "port").  One JSON line per clip length."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import bfm_ref as br, raster_ref as rr
from voicepuppet_amd.utils import reconstruct_mesh as vrm, mesh_core

fm = br.synthetic_facemodel(0, nlat=189, nlon=189, smooth=True)
model = vrm.DeviceFaceModel(fm)
N, F = model.nver, model.ntri
dev = model.device


def ev_time(fn, warm=3, steps=20):
  for _ in range(warm):
    fn()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  a.record()
  for _ in range(steps):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / steps


for T in (1, 25, 125):
  coeff, angles = br.synthetic_coeffs(T, 1)
  coeff_d = torch.from_numpy(coeff).to(dev)
  o = vrm.reconstruct_clip(coeff_d, model, angles, shared_texture=True, full=False)
  image = torch.zeros(T, 224, 224, 3, dtype=torch.uint8, device=dev)
  mask = torch.zeros(T, 224, 224, dtype=torch.uint8, device=dev)
  depth = torch.empty(T, 224, 224, dtype=torch.float32, device=dev)

  def raster():
    depth.fill_(-99999.0)
    mesh_core.render_colors(image, mask, o["vertices"], model.tri, o["colors"], depth)

  ms_rec = ev_time(lambda: vrm.reconstruct_clip(coeff_d, model, angles, shared_texture=True, full=False))
  ms_ras = ev_time(raster)
  # algorithmic bytes of the reconstruction: three float64 bases + means once per clip, float32 vertices + colours out
  alg = 8.0 * 3 * N * (80 + 64 + 80 + 2) + 4.0 * T * N * 6
  line = {"config": "BFM reconstruction + rasteriser, %d vertices / %d triangles, 224x224, clip of %d frames" % (N, F, T),
          "dtype": "f64 reconstruction, f32/u8 rasteriser", "reconstruct_ms": ms_rec, "raster_ms": ms_ras,
          "frames_per_s": T / ((ms_rec + ms_ras) * 1e-3), "coverage": float((mask > 0).float().mean()),
          "roofline": {"kernel": "bfm_linear_kernel+bfm_vertex_kernel", "bound": "hbm", "achieved": alg / (ms_rec * 1e-3) / 1e9, "peak": 8000.0,
                       "unit": "GB/s", "frac": alg / (ms_rec * 1e-3) / 1e9 / 8000.0, "traffic": None}}
  if T == 25:
    v, c = o["vertices"][0].cpu().numpy(), o["colors"][0].cpu().numpy()
    tri = model.tri.cpu().numpy()
    t0 = time.perf_counter(); want = br.reconstruction_rotation(coeff[:1], fm, angles[:1]); t_rec = time.perf_counter() - t0
    base = {"reconstruct": {"value": 1 / t_rec, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": "1 frame, numpy float64 restatement"}}
    if rr.have_compiled_reference():
      t0 = time.perf_counter(); ref = rr.render_colors_ref(v, tri, c, 224, 224); t_ras = time.perf_counter() - t0
      base["raster"] = {"value": 1 / t_ras, "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "1 frame, compiled mesh_core.cpp"}
      line["raster_bit_exact_vs_reference"] = bool(np.array_equal(ref[0], image[0].cpu().numpy()) and np.array_equal(ref[1], mask[0].cpu().numpy()))
    line["cpu_baseline"] = base
  print(json.dumps(line))
