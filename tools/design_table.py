#!/usr/bin/env python3
"""DESIGN.md section 3's kernel table from a per-kernel collection: python tools/design_table.py gpurun_out/<tag>_k/kernels.json
(prints the markdown rows; the commentary column is this script's)."""
import json
import sys

recs, ALL = {}, []
for x in json.load(open(sys.argv[1])):
    if x.get("case"):
        recs[x["case"].split(":")[0].split(" (")[0][:44]] = x
        ALL.append(x)


def g(key):
    if key in recs:
        return recs[key]
    hits = [v for k, v in recs.items() if k.startswith(key)]
    if len(hits) != 1:
        raise KeyError("%s: %d matches" % (key, len(hits)))
    return hits[0]


def us(k): return "%.1f" % g(k)["avg_us"]
def fr(k, d=3): return ("%." + str(d) + "f") % g(k)["frac_of_8TBps_at_avg"]
def tr(k, d=4): return ("%." + str(d) + "f") % g(k)["traffic_over_algorithmic"]
def vb(k): return "%.2f" % g(k)["sq"]["valu_active_wave_cycles_per_simd_cycle"]
def clk(k): return "%.2f" % g(k)["sq"]["shader_clock_GHz"]


M2S, M2SB = "map_ops metallic -> diffuse/specular 4096^2", "map_ops metallic -> diffuse/specular backwar"
s2m = sorted((r for r in ALL if r["kernel"].startswith("specular_to_metallic")), key=lambda r: r["avg_us"])
print(f'''| kernel (4096² probe shape unless noted) | algorithmic bytes per pixel | µs | frac | PMC traffic ÷ algorithmic | VALU active (wave-cycles per SIMD cycle; saturates at 1.55 plain fp32 / 0.91 packed) | bound |
|---|---|---|---|---|---|---|
| `cook_torrance_kernel<1,0,float,float,4,…>` — the bench workload | 32 in + 12 out | **{us('fwd_f32')}** | **{fr('fwd_f32')}** | {tr('fwd_f32')} | {vb('fwd_f32')} | HBM (0.99 of its bare access pattern, measured in the bench line) |
| same, fp16 maps → fp32, 4 materials (`…,__half,float,8,…`) | 16 + 12 | {us('fwd_f16')} | {fr('fwd_f16')} | {tr('fwd_f16')} | {vb('fwd_f16')} | HBM / VALU |
| same, fp16 → fp16 | 16 + 6 | {us('fwd_f16_f16')} | {fr('fwd_f16_f16')} | {tr('fwd_f16_f16')} | {vb('fwd_f16_f16')} | VALU (power-limited clock: {clk('fwd_f16_f16')} GHz) |
| `cook_torrance_repeat_kernel` — 2048² maps, `tile(2)` → 4096² | 8 (a quarter of 32) + 12 | **{us('tiled')}** | **{fr('tiled')}** | {tr('tiled')} | {vb('tiled')} | HBM (writes) |
| same, fp16 maps | 4 + 12 | {us('tiled_f16')} | {fr('tiled_f16')} | {tr('tiled_f16')} | {vb('tiled_f16')} | VALU |
| same, 4 point lights (`…, MULTI`; round 5) | 8 + 12 | {us('tiled_multi')} | {fr('tiled_multi')} | **{tr('tiled_multi')}** | {vb('tiled_multi')} | VALU |
| same, a THIN row band of the tiled image — rows [1536, 2560) of 4096: a quarter of the image across the period boundary, one of four ranks' shard of ONE tiled material (round 6: the walk's window of source rows; until then the wrap-around form at 1.40 × traffic) | 8 + 12 per band pixel | {us('tiled_band')} | {fr('tiled_band')} | **{tr('tiled_band')}** | {vb('tiled_band')} | HBM (a quarter of the launch: 4.2 M pixels) |
| `cook_torrance_batch_kernel<…,2,4,…>` — 4 materials, 16 lights, fp16 maps (config 5's share) | 16 + 12 | {us('fwd_16_lights')} | {fr('fwd_16_lights')} | {tr('fwd_16_lights')} | {vb('fwd_16_lights')} | **VALU**: 30.2 vector instructions per (pixel, light); at the {clk('fwd_16_lights')} GHz the counters measured under this launch ≈ 0.78 of issue |
| `cook_torrance_backward_kernel<1,0,4,…,float,…>` | 44 in + 32 out | {us('bwd_f32')} | {fr('bwd_f32')} | {tr('bwd_f32')} | {vb('bwd_f32')} | HBM |
| `cook_torrance_backward_stream_kernel<1,0,true>` — fp16 maps (directional: `<0,0,true>`) | 28 + 16 | {us('bwd_f16')} ({us('bwd_dir_f16')}) | {fr('bwd_f16')} ({fr('bwd_dir_f16')}) | 1.0001 | {vb('bwd_f16')} ({vb('bwd_dir_f16')}) | VALU issue |
| **`cook_torrance_repeat_backward_kernel<1,0,float,false>`** — folded gradient of 2048² maps under `tile(2)` → 4096², point light (rounds 5-6) | 12 per output pixel + 64 per texel (= 26.7 per output pixel) | **{us('tiled_bwd_f32')}** | {fr('tiled_bwd_f32')} | **{tr('tiled_bwd_f32')}** | {vb('tiled_bwd_f32')} | **VALU issue** (three waves per SIMD since round 6; round 5: 122.0 µs at two); the two-kernel form it replaces: 319-327 µs, 1.95 GB |
| same, ONE directional light (the repeats' upstream values summed before the chain rule; XCDs walk runs of 1 024 tiles) | as above | **{us('tiled_bwd_dir_f32')}** | **{fr('tiled_bwd_dir_f32')}** | {tr('tiled_bwd_dir_f32')} | {vb('tiled_bwd_dir_f32')} | HBM (8-byte streams) |
| same, fp16 maps | 12 + 32 per texel | {us('tiled_bwd_f16')} | {fr('tiled_bwd_f16')} | {tr('tiled_bwd_f16')} | {vb('tiled_bwd_f16')} | VALU issue (round 5: 127.3) |
| same with the loss policy — the rendering-loss step over tiled maps (fp32 / fp16 maps) | as above | {us('tiled_bwd_loss_f32')} / {us('tiled_bwd_loss_f16')} | {fr('tiled_bwd_loss_f32')} / {fr('tiled_bwd_loss_f16')} | {tr('tiled_bwd_loss_f32', 3)} / {tr('tiled_bwd_loss_f16', 3)} | {vb('tiled_bwd_loss_f32')} / {vb('tiled_bwd_loss_f16')} | VALU issue (round 5: 135.7 / 141.6) |
| same, 4 point lights (`…, MULTI`) | as above | {us('tiled_multi_bwd')} | {fr('tiled_multi_bwd')} | {tr('tiled_multi_bwd')} | {vb('tiled_multi_bwd')} | VALU (two passes over the lights per position; round 5: 364.6) |
| `cook_torrance_mse_step_kernel` — rendering-loss step, fp32 | 44 + 32 | {us('loss_step_f32')} | {fr('loss_step_f32')} | {tr('loss_step_f32')} | {vb('loss_step_f32')} | HBM / VALU |
| `cook_torrance_mse_stream_kernel` — the same, fp16 maps | 28 + 16 | {us('loss_step_f16')} | {fr('loss_step_f16')} | {tr('loss_step_f16')} | {vb('loss_step_f16')} | VALU issue |
| `cook_torrance_blend_kernel` — blend + re-decode + render | 68 + 12 | {us('blend_fused')} | {fr('blend_fused')} | {tr('blend_fused')} | {vb('blend_fused')} | HBM (20 streams) |
| **`cook_torrance_repeat_blend_kernel`** — the fused blend over TILED maps (round 6): 2 × 2048² materials + mask under `tile(2)` → 4096², blended once per texel, evaluated at every repeat | 68 per texel + 12 per output pixel | **{us('blend_tiled_fwd')}** | {fr('blend_tiled_fwd')} | **{tr('blend_tiled_fwd')}** | {vb('blend_tiled_fwd')} | HBM / VALU (until round 6 the wrap-around form, the blend at every output pixel: 154 µs) |
| `cook_torrance_blend_backward_kernel` | 80 + 68 | {us('blend_bwd')} | {fr('blend_bwd')} | {tr('blend_bwd')} | {vb('blend_bwd')} | HBM (37 streams) |
| **`cook_torrance_repeat_blend_backward_kernel`** — the fused blend's backward over TILED maps (round 6): 2 × 2048² materials + mask under `tile(2)` → 4096² | 12 per output pixel + 136 per texel | **{us('blend_bwd_tiled')}** | {fr('blend_bwd_tiled')} | **{tr('blend_bwd_tiled')}** | {vb('blend_bwd_tiled')} | VALU issue (two waves per SIMD: 254 registers); unfused: blend backward + decode backward + folded render backward |
| `colour_kernel` / `colour_backward_kernel` | 12 + 12 / 24 + 12 | {us('map_ops srgb_to_linear 3 x 4096^2 fp32')} / {us('map_ops srgb_to_linear backward')} | {fr('map_ops srgb_to_linear 3 x 4096^2 fp32', 2)} / {fr('map_ops srgb_to_linear backward', 2)} | 1.0001 | 0.25 | HBM |
| `metallic_to_specular_kernel` / its backward | 16 + 24 / 40 + 16 | {us(M2S)} / {us(M2SB)} | {fr(M2S, 2)} / {fr(M2SB, 2)} | 1.0002 | 0.21-0.29 | HBM |
| `specular_to_metallic_kernel` / its backward | 24 + 16 / 48 + 24 | {s2m[0]['avg_us']:.1f} / {s2m[1]['avg_us']:.1f} | {s2m[0]['frac_of_8TBps_at_avg']:.2f} / {s2m[1]['frac_of_8TBps_at_avg']:.2f} | 1.0001 | 0.26-0.33 | HBM |
| `blend_kernel<false / true>`, `sigmoid_mask_kernel` | 28 + 12; 8 + 4 | {us('blend_maps 3 ch')} / {us('blend_maps normals')}; {us('sigmoid mask')} | {fr('blend_maps 3 ch', 2)} / {fr('blend_maps normals', 2)}; {fr('sigmoid mask', 2)} | 1.0001 | 0.05-0.18 | HBM |
| `decode_normal_kernel` (in place) | 12 + 12 | {us('map_ops decode_normal in place')} | {fr('map_ops decode_normal in place', 2)} | 1.0001 | 0.07 | HBM |
| `unpack_dense_kernel<uint8,3,…>` — an RGB image's samples → 3 float32 planes (a normal map: decoded in the same pass) | 3 + 12 | {us('unpack_image 4096^2 RGB uint8 samples')} ({us('unpack_image 4096^2 RGB uint8 normal')}) | {fr('unpack_image 4096^2 RGB uint8 samples', 2)} ({fr('unpack_image 4096^2 RGB uint8 normal', 2)}) | 1.0001 | 0.41 (0.54) | HBM (writes) |
| **`resize_down_kernel<S,…>`** — whole-factor down-scale, register-only band walk (round 5): ONE 3-plane 4096² map (201 MB: it stays in the 256 MB memory-side cache between launches) → 2048² / 1024² / 512² | 4 per input + 4 per output pixel | {us('resize 3 x 4096^2 -> 2048')} / {us('resize 3 x 4096^2 -> 1024')} / {us('resize 3 x 4096^2 -> 512')} | {fr('resize 3 x 4096^2 -> 2048', 2)} / {fr('resize 3 x 4096^2 -> 1024', 2)} / {fr('resize 3 x 4096^2 -> 512', 2)} | {tr('resize 3 x 4096^2 -> 2048', 3)} / {tr('resize 3 x 4096^2 -> 1024', 3)} / {tr('resize 3 x 4096^2 -> 512', 3)} | {vb('resize 3 x 4096^2 -> 2048')} | HBM + memory-side cache (round 4's strip kernel: 44.8 / 36.0; 8 ×: two passes, 185) |
| same, 8 planes (537 MB: nothing survives a launch; 2 × and 4 ×: lanes of 16 bytes, non-temporal loads) | same | {us('resize 8 x 4096^2 -> 2048')} / {us('resize 8 x 4096^2 -> 1024')} / {us('resize 8 x 4096^2 -> 512')} | {fr('resize 8 x 4096^2 -> 2048', 2)} / {fr('resize 8 x 4096^2 -> 1024', 2)} / {fr('resize 8 x 4096^2 -> 512', 2)} | {tr('resize 8 x 4096^2 -> 2048', 3)} / {tr('resize 8 x 4096^2 -> 1024', 3)} / {tr('resize 8 x 4096^2 -> 512', 3)} | {vb('resize 8 x 4096^2 -> 2048')} | HBM: what its bare pattern streams at (`r06_membench_resize.txt`) |
| `resize_strip_kernel` — other down-scales below 7 ×: 4096² → 1365², 3 / 8 planes | same | {us('resize 3 x 4096^2 -> 1365')} / {us('resize 8 x 4096^2 -> 1365')} | {fr('resize 3 x 4096^2 -> 1365', 2)} / {fr('resize 8 x 4096^2 -> 1365', 2)} | {tr('resize 3 x 4096^2 -> 1365', 3)} / {tr('resize 8 x 4096^2 -> 1365', 3)} | {vb('resize 3 x 4096^2 -> 1365')} | cached loads (read-only ceiling 5.6 TB/s = 0.70); three barrier-separated phases per tile |
| **`resize_stream_kernel`** (round 6) — a walk down the INPUT rows, every row read once, two-wave workgroups (walk \| width pass + stores): antialiased down-scales that are not a whole factor, from 7 × up (17 … 36 taps; until round 6 the strip kernel's WIDE instantiation: 44 / 130 µs, 0.58 / 0.52 at 1.15 × traffic): 4096² → 400², 3 / 8 planes | same | **{us('resize 3 x 4096^2 -> 400')} / {us('resize 8 x 4096^2 -> 400')}** (+ its tables kernel: 5 µs) | **{fr('resize 3 x 4096^2 -> 400', 2)} / {fr('resize 8 x 4096^2 -> 400', 2)}** | {tr('resize 3 x 4096^2 -> 400', 3)} / {tr('resize 8 x 4096^2 -> 400', 3)} | {vb('resize 3 x 4096^2 -> 400')} | HBM: the bare patterns of the same box in `r06_membench_resize.txt` (read-only 8 planes; 105:1 over 8 planes) |
| same kernel where it is NOT the rule (knob value 2): 4096² → 1365², 3 / 8 planes — its stores decide there: ahead of the strip kernel at boost clocks, level or behind at settled ones (§3.6, §9 #3) | same | {us('resize_walk3')} / {us('resize_walk8')} | {fr('resize_walk3', 2)} / {fr('resize_walk8', 2)} | {tr('resize_walk3', 3)} / {tr('resize_walk8', 3)} | {vb('resize_walk3')} | the result's stores (0.45 µs per MB against the strip kernel's 0.26) |
| `resize_up2_kernel<8>` 3 × 4096² → 6144² | same | {us('resize 3 x 4096^2 -> 6144')} | {fr('resize 3 x 4096^2 -> 6144', 2)} | {tr('resize 3 x 4096^2 -> 6144', 3)} | {vb('resize 3 x 4096^2 -> 6144')} (round 4: 0.69) | writes |
| `resize_down_kernel<2,4,2,3,true>` with the transposed two-tap weights — gradient of a 2× up-scale, 6 × 4096² upstream → 2048² (round 4's two-tap transpose: 0.71 at 1.17 × the bytes) | 4 per upstream + 4 per gradient pixel | {us('resize backward 6 x 4096^2')} | {fr('resize backward 6 x 4096^2', 2)} | {tr('resize backward 6 x 4096^2', 3)} | {vb('resize backward 6 x 4096^2')} | HBM |
| `resize_up2_backward_kernel<8,4>` — gradient of a 1.5× up-scale, 3 × 6144² upstream → 4096² | same | {us('resize backward 3 x 6144^2')} | {fr('resize backward 3 x 6144^2', 2)} | {tr('resize backward 3 x 6144^2', 3)} | {vb('resize backward 3 x 6144^2')} | VALU (its column products) / cached loads |
| `resize_backward_gather_kernel<8,8,true>` — gradient of a 2× down-scale, 3 × 2048² upstream → 4096² (+ 5 µs of tables kernel per call) | same | {us('resize backward 3 x 2048^2')} | {fr('resize backward 3 x 2048^2', 2)} | {tr('resize backward 3 x 2048^2', 3)} | {vb('resize backward 3 x 2048^2')} | HBM (writes) |''')
