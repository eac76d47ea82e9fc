"""Oracle-compared GPU cases for the MFMA conv family, shared by tests/test_gpu_conv.py (runs them on the GPU against
torch-CPU fp32) and tests/test_conv_dispatch.py (CPU: proves, through the dispatcher's dry-run entry points, that every
kernel instantiation the BASELINE train plans launch is hit by one of these cases).

Shapes are the reference network's own layer shapes (Keras ResNet50 v1 to conv4_block6_out, reference
models/feature_extractor.py:4-11; RPN / Dense heads, models/detectors/*.py) at 375x1242: the dispatch heuristics depend on
M = batch * Ho * Wo, so the cases keep batch 4 where the tile choice needs it and shrink the batch where it does not.
"""

# ---- forward convolutions: n, h, w (input grid), cin, cout, k, stride, pad, epilogue flags
FPROP = [
    # 128 x 128 tiles
    dict(id="c4_1x1_256_1024_stats", n=4, h=24, w=78, cin=256, cout=1024, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="c4_s2_512_1024_stats", n=4, h=47, w=156, cin=512, cout=1024, k=1, s=2, p=0, bias=True, relu=False, stats=True),
    # kw-sharing 3x3 (>= 160 m-tiles); Wo = 311 is prime, M = 29234 is not a multiple of 128
    dict(id="c2_3x3_64_64_relu", n=1, h=94, w=311, cin=64, cout=64, k=3, s=1, p=1, bias=True, relu=True, stats=False),
    dict(id="c2_3x3_64_64_stats", n=1, h=94, w=311, cin=64, cout=64, k=3, s=1, p=1, bias=True, relu=False, stats=True),
    dict(id="c3_3x3_128_128_stats", n=3, h=47, w=156, cin=128, cout=128, k=3, s=1, p=1, bias=True, relu=False, stats=True),
    # tile runs (short K)
    dict(id="c2_1x1_64_256_stats_run4", n=4, h=94, w=311, cin=64, cout=256, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="c3_s2_256_128_stats_run2", n=6, h=94, w=311, cin=256, cout=128, k=1, s=2, p=0, bias=True, relu=False, stats=True),    # strided A rows in a run (344 m-tiles)
    dict(id="c3_s2_256_512_stats", n=4, h=94, w=311, cin=256, cout=512, k=1, s=2, p=0, bias=True, relu=False, stats=True),         # 4 slices, 512 channels: one 128 x 128 tile
    # one 128 x 64 tile per workgroup at the benchmark's sizes
    dict(id="c2_1x1_256_64_stats", n=4, h=94, w=311, cin=256, cout=64, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="c4_1x1_1024_256_stats", n=4, h=24, w=78, cin=1024, cout=256, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="c4_3x3_256_256_stats", n=4, h=24, w=78, cin=256, cout=256, k=3, s=1, p=1, bias=True, relu=False, stats=True),
    dict(id="rpn_3x3_1024_256_relu", n=1, h=24, w=78, cin=1024, cout=256, k=3, s=1, p=1, bias=True, relu=True, stats=False),
    # patch-resident 3x3 kernel (conv3x3_patch: cin >= 256, every 8 x 16 pixel x 64 channel workgroup on a CU of its own): the plans' conv4 /
    # RPN launches (c4_3x3_256_256_stats, c4_3x3_256_256_b2 and rpn_3x3_1024_256_relu above), 16 channel chunks at batch 4, and grids that end
    # inside tiles in both directions (13 x 21: one full and one 5-wide tile column, rows 8..12 of the second tile row)
    dict(id="rpn_3x3_1024_256_relu_patch_b4", n=4, h=24, w=78, cin=1024, cout=256, k=3, s=1, p=1, bias=True, relu=True, stats=False, ws=True),   # the plan's launch: 240 workgroups (the workspace the model attaches is ignored)
    dict(id="patch_3x3_256_128_stats_ragged", n=3, h=13, w=21, cin=256, cout=128, k=3, s=1, p=1, bias=True, relu=False, stats=True),
    dict(id="patch_3x3_320_64_nobias_ragged", n=1, h=9, w=33, cin=320, cout=64, k=3, s=1, p=1, bias=False, relu=False, stats=False),     # five chunks: odd count
    # weights-resident 3x3 kernel (conv3x3_wres: 64 input channels, at least two 8 x 16 pixel tiles per CU): conv2's layers at the
    # benchmark's batch (240 workgroups x 4 tiles), two channel parts with a ragged grid and an uneven share (270 tiles on 128 x 2 workgroups)
    dict(id="wres_3x3_64_64_stats_b4", n=4, h=94, w=311, cin=64, cout=64, k=3, s=1, p=1, bias=True, relu=False, stats=True),
    dict(id="wres_3x3_64_128_relu_ragged", n=9, h=45, w=70, cin=64, cout=128, k=3, s=1, p=1, bias=True, relu=True, stats=False),
    dict(id="wres_3x3_64_128_stats_ragged", n=9, h=45, w=70, cin=64, cout=128, k=3, s=1, p=1, bias=False, relu=False, stats=True),
    # feature-pyramid neck (BASELINE configs[4]): bias-only epilogues on shapes the C4 plans run with statistics
    dict(id="fpn_lateral4_1x1_1024_256_bias", n=4, h=24, w=78, cin=1024, cout=256, k=1, s=1, p=0, bias=True, relu=False, stats=False),
    dict(id="fpn_lateral3_1x1_512_256_bias", n=4, h=47, w=156, cin=512, cout=256, k=1, s=1, p=0, bias=True, relu=False, stats=False),       # 460 tiles: 128 x 128
    dict(id="fpn_p2_3x3_256_256_bias_relu", n=1, h=94, w=311, cin=256, cout=256, k=3, s=1, p=1, bias=True, relu=True, stats=False),        # 256 output channels: plain 128 x 128 tiles, no kw sharing
    dict(id="fpn_output4_3x3_256_256_bias_b8", n=8, h=24, w=78, cin=256, cout=256, k=3, s=1, p=1, bias=True, relu=False, stats=False),
    # short-K 1x1 layers without statistics on the streaming kernel (conv1x1_stream: K = 64 / 128, M >= 4096)
    dict(id="stream_64_64_bias", n=4, h=94, w=311, cin=64, cout=64, k=1, s=1, p=0, bias=True, relu=False, stats=False),
    dict(id="stream_128_512_bias", n=4, h=47, w=156, cin=128, cout=512, k=1, s=1, p=0, bias=True, relu=False, stats=False),
    dict(id="c3_1x1_128_512_stats", n=4, h=47, w=156, cin=128, cout=512, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="stream_64_256_relu_tail", n=1, h=67, w=63, cin=64, cout=256, k=1, s=1, p=0, bias=True, relu=True, stats=False),      # M = 4221: 13-pixel tail group
    dict(id="stream_128_64_nobias", n=1, h=65, w=64, cin=128, cout=64, k=1, s=1, p=0, bias=False, relu=False, stats=False),
    # small shapes (tails, odd grids)
    dict(id="small_1x1_stats", n=2, h=13, w=17, cin=64, cout=256, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="small_3x3_relu", n=2, h=12, w=10, cin=64, cout=64, k=3, s=1, p=1, bias=True, relu=True, stats=False),
    dict(id="small_s2_stats", n=1, h=15, w=21, cin=256, cout=128, k=1, s=2, p=0, bias=False, relu=False, stats=True),
    dict(id="small_run_512", n=4, h=64, w=64, cin=64, cout=512, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="small_3x3_128_stats", n=1, h=9, w=11, cin=128, cout=128, k=3, s=1, p=1, bias=True, relu=False, stats=True),
    dict(id="c4_3x3_256_256_b2", n=2, h=24, w=78, cin=256, cout=256, k=3, s=1, p=1, bias=True, relu=False, stats=True),
    # 64-row tiles (round 5): long K on a grid that 128-row tiles would leave half empty -- conv4's 1024 -> 256 and stride-2 512 -> 256 at
    # M = 3,744 (ResNet-101, batch 2: BASELINE configs[3]), and a 58-row tail
    dict(id="c4_1x1_1024_256_stats_b2_bm64", n=2, h=24, w=78, cin=1024, cout=256, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="c4_s2_512_256_stats_b2_bm64", n=2, h=47, w=156, cin=512, cout=256, k=1, s=2, p=0, bias=True, relu=False, stats=True),
    dict(id="bm64_1x1_512_128_relu_tail", n=1, h=13, w=34, cin=512, cout=128, k=1, s=1, p=0, bias=True, relu=True, stats=False),
    # split-K fix-up form (ws: the descriptor carries a workspace): fewer tiles than CUs, >= 64 K slices
    dict(id="rpn_like_3x3_1024_256_relu_fix_w15", n=4, h=24, w=15, cin=1024, cout=256, k=3, s=1, p=1, bias=True, relu=True, stats=False, ws=True),   # (rows narrower than a 16-pixel patch tile: the tile kernel's pair form)
    dict(id="k4096_1x1_stats_fix", n=1, h=24, w=78, cin=4096, cout=256, k=1, s=1, p=0, bias=True, relu=False, stats=True, ws=True),
    dict(id="k4096_1x1_relu_fix", n=1, h=12, w=10, cin=4096, cout=64, k=1, s=1, p=0, bias=True, relu=True, stats=False, ws=True),
    dict(id="small_3x3_576_oddk_stats_fix", n=1, h=12, w=10, cin=576, cout=64, k=3, s=1, p=1, bias=True, relu=False, stats=True, ws=True),   # 81 slices: 41 + 40, one tile, 14 idle workgroups
]

# ---- fp8 (e4m3) forward convolutions (frcnn_conv2d_fprop_fp8): the layers of the same network whose cin is a multiple of 128
FPROP_FP8 = [
    dict(id="f8_c4_1x1_256_1024_stats", n=4, h=24, w=78, cin=256, cout=1024, k=1, s=1, p=0, bias=True, relu=False, stats=True),       # 128 x 128 tiles, 2 slices
    dict(id="f8_c4_s2_512_1024_stats", n=4, h=47, w=156, cin=512, cout=1024, k=1, s=2, p=0, bias=True, relu=False, stats=True),
    dict(id="f8_c3_3x3_128_128_stats", n=3, h=47, w=156, cin=128, cout=128, k=3, s=1, p=1, bias=True, relu=False, stats=True),       # kw-sharing
    dict(id="f8_c3_1x1_128_512_stats_run", n=4, h=47, w=156, cin=128, cout=512, k=1, s=1, p=0, bias=True, relu=False, stats=True),   # ONE slice per tile
    dict(id="f8_c3_1x1_512_128_stats_b2", n=2, h=47, w=156, cin=512, cout=128, k=1, s=1, p=0, bias=True, relu=False, stats=True),     # 115 m-tiles: 128 x 64, three workgroups per CU
    dict(id="f8_c3_1x1_512_128_stats", n=4, h=47, w=156, cin=512, cout=128, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="f8_c3_s2_256_512_stats", n=4, h=94, w=311, cin=256, cout=512, k=1, s=2, p=0, bias=True, relu=False, stats=True),
    dict(id="f8_c3_s2_256_128_stats", n=4, h=94, w=311, cin=256, cout=128, k=1, s=2, p=0, bias=True, relu=False, stats=True),
    dict(id="f8_c2_1x1_256_64_stats", n=4, h=94, w=311, cin=256, cout=64, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="f8_c4_1x1_1024_256_stats", n=4, h=24, w=78, cin=1024, cout=256, k=1, s=1, p=0, bias=True, relu=False, stats=True),
    dict(id="f8_c4_s2_512_256_stats", n=4, h=47, w=156, cin=512, cout=256, k=1, s=2, p=0, bias=True, relu=False, stats=True),
    dict(id="f8_c4_3x3_256_256_stats", n=4, h=24, w=78, cin=256, cout=256, k=3, s=1, p=1, bias=True, relu=False, stats=True),
    dict(id="f8_rpn_3x3_1024_256_relu", n=1, h=24, w=78, cin=1024, cout=256, k=3, s=1, p=1, bias=True, relu=True, stats=False),
    dict(id="f8_rpn_3x3_1024_256_relu_fix_b4", n=4, h=24, w=78, cin=1024, cout=256, k=3, s=1, p=1, bias=True, relu=True, stats=False, ws=True),
    dict(id="f8_rpn_3x3_1024_256_relu_b8", n=8, h=24, w=78, cin=1024, cout=256, k=3, s=1, p=1, bias=True, relu=True, stats=False),    # configs[4]'s batch: 128 x 128 tiles
    dict(id="f8_fpn_p2_3x3_256_256_bias", n=1, h=94, w=311, cin=256, cout=256, k=3, s=1, p=1, bias=True, relu=False, stats=False),         # pyramid level 2: 128 x 128 tiles, no statistics
    # the pyramid's lateral 1x1 convolutions on the backbone's twins: bias only, no statistics
    dict(id="f8_fpn_lateral4_1024_256_bias", n=8, h=24, w=78, cin=1024, cout=256, k=1, s=1, p=0, bias=True, relu=False, stats=False),
    dict(id="f8_fpn_lateral3_512_256_bias_run", n=8, h=47, w=156, cin=512, cout=256, k=1, s=1, p=0, bias=True, relu=False, stats=False),
    dict(id="f8_small_s2_stats", n=1, h=15, w=21, cin=256, cout=128, k=1, s=2, p=0, bias=False, relu=False, stats=True),
    dict(id="f8_small_3x3_128_stats", n=1, h=9, w=11, cin=128, cout=128, k=3, s=1, p=1, bias=True, relu=False, stats=True),
    dict(id="f8_small_1x1_relu", n=2, h=13, w=17, cin=384, cout=72, k=1, s=1, p=0, bias=True, relu=True, stats=False),
]

# ---- fp8 data gradients (frcnn_conv2d_dgrad_fp8: e5m2 gradient x e4m3 transposed weights); fields as DGRAD below
DGRAD_FP8 = [
    dict(id="f8_rpn_dg_3x3_256_1024_res_red", n=4, h=24, w=78, cin=256, cout=1024, k=3, res=True, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="f8_c4_dg_1024_256_red", n=4, h=24, w=78, cin=1024, cout=256, k=1, res=False, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="f8_c4_dg_3x3_256_256_red", n=4, h=24, w=78, cin=256, cout=256, k=3, res=False, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="f8_c4_dg_256_1024_res_mask_red", n=4, h=24, w=78, cin=256, cout=1024, k=1, res=True, res_mask=True, red=True, mask=True, scatter=1),
    dict(id="f8_c4_dg_s2_256_512_scatter_plain_b2", n=2, h=24, w=78, cin=256, cout=512, k=1, res=False, res_mask=False, red=False, mask=False, scatter=2),
    dict(id="f8_c4_dg_s2_256_512_scatter_plain", n=4, h=24, w=78, cin=256, cout=512, k=1, res=False, res_mask=False, red=False, mask=False, scatter=2),
    dict(id="f8_c3_dg_3x3_128_128_red", n=3, h=47, w=156, cin=128, cout=128, k=3, res=False, res_mask=False, red=True, mask=True, scatter=1),      # kw-sharing
    dict(id="f8_c3_dg_128_512_res_mask_red_run", n=4, h=47, w=156, cin=128, cout=512, k=1, res=True, res_mask=True, red=True, mask=True, scatter=1),
    dict(id="f8_c2_dg_256_64_red", n=4, h=94, w=311, cin=256, cout=64, k=1, res=False, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="f8_c2_dg_256_64_res_plain", n=4, h=94, w=311, cin=256, cout=64, k=1, res=True, res_mask=False, red=False, mask=False, scatter=1),
    dict(id="f8_c4_dg_1024_256_red_b8", n=8, h=24, w=78, cin=1024, cout=256, k=1, res=False, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="f8_c4_dg_256_1024_res_mask_red_b8_run", n=8, h=24, w=78, cin=256, cout=1024, k=1, res=True, res_mask=True, red=True, mask=True, scatter=1),
    dict(id="f8_c3_dg_s2_128_256_scatter_plain_b8_run", n=8, h=47, w=156, cin=128, cout=256, k=1, res=False, res_mask=False, red=False, mask=False, scatter=2),
    # the pyramid's 3x3 convolutions (no BatchNorm behind them: plain data gradients, with or without the RoI branch's gradient as residual)
    dict(id="f8_fpn_dg_3x3_256_256_plain_b8", n=8, h=24, w=78, cin=256, cout=256, k=3, res=False, res_mask=False, red=False, mask=False, scatter=1),
    dict(id="f8_fpn_dg_3x3_256_256_plain_p5", n=8, h=12, w=39, cin=256, cout=256, k=3, res=False, res_mask=False, red=False, mask=False, scatter=1),
    dict(id="f8_fpn_dg_p2_3x3_256_256_res", n=1, h=94, w=311, cin=256, cout=256, k=3, res=True, res_mask=False, red=False, mask=False, scatter=1),
    dict(id="f8_c4_dg_s2_1024_512_scatter_res_red_b8", n=8, h=24, w=78, cin=1024, cout=512, k=1, res=True, res_mask=False, red=True, mask=True, scatter=2),
    dict(id="f8_c4_dg_s2_1024_512_scatter_res_plain_b8", n=8, h=24, w=78, cin=1024, cout=512, k=1, res=True, res_mask=False, red=False, mask=False, scatter=2),
    dict(id="f8_small_dg_scatter_res_red", n=2, h=12, w=39, cin=512, cout=256, k=1, res=True, res_mask=False, red=True, mask=True, scatter=2),
    dict(id="f8_small_dg_3x3_nomask", n=2, h=13, w=17, cin=128, cout=128, k=3, res=False, res_mask=False, red=True, mask=False, scatter=1),
]

# ---- data gradients: dz grid n x h x w with cin channels -> gx with cout channels; k = 1 or 3 (stride 1, pad k//2);
# scatter 2: the gradient of a stride-2 1x1 convolution, written to every second pixel of a 2h x 2w (-1) grid.
# res: residual added (ADD_RES); res_mask: bit mask on the residual; red: fused BatchNorm-backward reduce (mask: with ReLU bits)
DGRAD = [
    # 128 x 128 tiles
    dict(id="c3_dg_512_128_red", n=4, h=47, w=156, cin=512, cout=128, k=1, res=False, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="c3_dg_128_512_res_red", n=2, h=47, w=156, cin=128, cout=512, k=1, res=True, res_mask=True, red=True, mask=True, scatter=1),
    dict(id="rpn_dg_3x3_256_1024_res_red", n=4, h=24, w=78, cin=256, cout=1024, k=3, res=True, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="c3_dg_s2_128_256_scatter", n=4, h=47, w=156, cin=128, cout=256, k=1, res=False, res_mask=False, red=False, mask=False, scatter=2),
    dict(id="c3_dg_s2_512_256_scatter_res_red", n=4, h=47, w=156, cin=512, cout=256, k=1, res=True, res_mask=False, red=True, mask=True, scatter=2),
    # kw-sharing 3x3
    dict(id="c2_dg_3x3_64_64_red", n=1, h=94, w=311, cin=64, cout=64, k=3, res=False, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="c3_dg_3x3_128_128_red", n=3, h=47, w=156, cin=128, cout=128, k=3, res=False, res_mask=False, red=True, mask=True, scatter=1),
    # tile runs
    dict(id="c2_dg_64_256_res_red_run4", n=4, h=94, w=311, cin=64, cout=256, k=1, res=True, res_mask=True, red=True, mask=True, scatter=1),
    dict(id="c2_dg_64_256_plain_run", n=4, h=94, w=311, cin=64, cout=256, k=1, res=False, res_mask=False, red=False, mask=False, scatter=1),
    # one 128 x 64 tile per workgroup
    dict(id="c2_dg_256_64_red", n=4, h=94, w=311, cin=256, cout=64, k=1, res=False, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="c2_dg_64_64_plain", n=4, h=94, w=311, cin=64, cout=64, k=1, res=False, res_mask=False, red=False, mask=False, scatter=1),
    dict(id="c4_dg_1024_256_red", n=4, h=24, w=78, cin=1024, cout=256, k=1, res=False, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="c4_dg_1024_256_red_b2_bm64", n=2, h=24, w=78, cin=1024, cout=256, k=1, res=False, res_mask=False, red=True, mask=True, scatter=1),   # 64-row tiles (ResNet-101, batch 2)
    dict(id="c4_dg_3x3_256_256_red", n=4, h=24, w=78, cin=256, cout=256, k=3, res=False, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="rpn_heads_dg_128_256", n=4, h=24, w=78, cin=128, cout=256, k=1, res=False, res_mask=False, red=False, mask=False, scatter=1),
    dict(id="rpn_heads_dg_128_256_b8", n=8, h=24, w=78, cin=128, cout=256, k=1, res=False, res_mask=False, red=False, mask=False, scatter=1),   # 128 x 128 tiles
    dict(id="c4_dg_s2_256_512_scatter_plain", n=4, h=24, w=78, cin=256, cout=512, k=1, res=False, res_mask=False, red=False, mask=False, scatter=2),
    dict(id="fpn_dg_p2_3x3_256_256_res", n=1, h=94, w=311, cin=256, cout=256, k=3, res=True, res_mask=False, red=False, mask=False, scatter=1),   # plain 128 x 128 tiles
    # small shapes
    dict(id="small_dg_run", n=2, h=12, w=39, cin=256, cout=64, k=1, res=True, res_mask=True, red=True, mask=True, scatter=1),
    dict(id="small_dg_3x3", n=2, h=13, w=17, cin=128, cout=128, k=3, res=False, res_mask=False, red=True, mask=True, scatter=1),
    dict(id="wres_dg_3x3_64_64_red_b4", n=4, h=94, w=311, cin=64, cout=64, k=3, res=False, res_mask=False, red=True, mask=True, scatter=1),                # conv3x3_wres, SMODE 2
    dict(id="wres_dg_3x3_64_64_plain_ragged", n=7, h=90, w=100, cin=64, cout=64, k=3, res=False, res_mask=False, red=False, mask=False, scatter=1),   # 588 tiles: shares of 3 and 2
    dict(id="wres_dg_3x3_64_64_red_nomask_ragged", n=7, h=90, w=100, cin=64, cout=64, k=3, res=False, res_mask=False, red=True, mask=False, scatter=1),
    dict(id="patch128_rpn_dg_3x3_256_1024_plain_b4", n=4, h=24, w=78, cin=256, cout=1024, k=3, res=False, res_mask=False, red=False, mask=False, scatter=1),   # conv3x3_patch, 128 channels per workgroup: the plan's RPN data gradient (480 workgroups)
    dict(id="patch128_dg_3x3_128_512_red_ragged", n=12, h=13, w=21, cin=128, cout=512, k=3, res=False, res_mask=False, red=True, mask=True, scatter=1),   # 48 ragged tiles x 4 parts of 128
    dict(id="patch_dg_3x3_256_256_red_ragged", n=2, h=13, w=21, cin=256, cout=256, k=3, res=False, res_mask=False, red=True, mask=True, scatter=1),   # conv3x3_patch, SMODE 2, partial tiles
    dict(id="patch_dg_3x3_256_64_red_nomask", n=1, h=24, w=78, cin=256, cout=64, k=3, res=False, res_mask=False, red=True, mask=False, scatter=1),
    dict(id="patch_dg_3x3_512_128_plain", n=1, h=17, w=40, cin=512, cout=128, k=3, res=False, res_mask=False, red=False, mask=False, scatter=1),
    dict(id="small_dg_nomask", n=1, h=24, w=78, cin=1024, cout=256, k=1, res=False, res_mask=False, red=True, mask=False, scatter=1),
    dict(id="small_dg_scatter", n=2, h=12, w=39, cin=512, cout=256, k=1, res=True, res_mask=False, red=True, mask=True, scatter=2),
    # split-K fix-up form (>= 64 K slices on fewer tiles than CUs)
    dict(id="k4096_dg_red_fix", n=1, h=24, w=78, cin=4096, cout=256, k=1, res=False, res_mask=False, red=True, mask=True, scatter=1, ws=True),
    dict(id="k4096_dg_res_mask_red_fix", n=1, h=12, w=39, cin=4096, cout=64, k=1, res=True, res_mask=True, red=True, mask=True, scatter=1, ws=True),
    dict(id="k4096_dg_scatter_fix", n=1, h=12, w=39, cin=4096, cout=64, k=1, res=True, res_mask=False, red=True, mask=True, scatter=2, ws=True),
    dict(id="dg_3x3_512_red_fix", n=1, h=13, w=17, cin=512, cout=64, k=3, res=False, res_mask=False, red=True, mask=True, scatter=1, ws=True),
    dict(id="dg_3x3_512_plain_fix", n=1, h=13, w=17, cin=512, cout=64, k=3, res=False, res_mask=False, red=False, mask=False, scatter=1, ws=True),
]

# ---- fp32-output / split-K GEMMs (RPN heads, Dense heads)
F32 = [
    dict(id="rpn_heads_f32", m=4 * 24 * 78, k=256, cout=128, split=1, bias=True),
    dict(id="dense_heads_splitk", m=1200, k=50176, cout=64, split=8, bias=False),
    dict(id="small_f32", m=300, k=6400, cout=64, split=1, bias=True),
    dict(id="small_splitk", m=300, k=6400, cout=64, split=8, bias=False),
]

# ---- weight gradients (single launches)
WGRAD = [
    dict(id="rpn_3x3_1024_256", n=1, h=24, w=78, cin=1024, cout=256, k=3, s=1, p=1),            # general addressing
    dict(id="rpn_heads_256_128", n=4, h=24, w=78, cin=256, cout=128, k=1, s=1, p=0),            # linear
    dict(id="small_3x3", n=2, h=12, w=10, cin=64, cout=64, k=3, s=1, p=1),
    dict(id="small_s2", n=1, h=15, w=21, cin=256, cout=128, k=1, s=2, p=0),
    dict(id="small_1x1", n=2, h=9, w=13, cin=128, cout=256, k=1, s=1, p=0),
    dict(id="c4_3x3", n=3, h=24, w=78, cin=256, cout=256, k=3, s=1, p=1),
    dict(id="c2_1x1", n=2, h=30, w=40, cin=64, cout=256, k=1, s=1, p=0),
    dict(id="fpn_p2_3x3_wide", n=1, h=94, w=311, cin=256, cout=256, k=3, s=1, p=1),             # M >= 24576, multi-tap: 128 x 128 tiles
    # more 3x3 shapes (added with the all-taps kernel of round 4, a sweep-build variant that was measured slower and is not dispatched):
    # ragged grids, conv2's shape at the benchmark's batch, conv3's
    dict(id="w3_ragged_64_128", n=2, h=13, w=37, cin=64, cout=128, k=3, s=1, p=1),
    dict(id="w3_c2_64_64_b4", n=4, h=94, w=311, cin=64, cout=64, k=3, s=1, p=1),
    dict(id="w3_c3_128_128", n=2, h=47, w=156, cin=128, cout=128, k=3, s=1, p=1),
    dict(id="small_3x3_general_cin32", n=2, h=12, w=10, cin=32, cout=64, k=3, s=1, p=1),       # (cin % 64 != 0: the per-tap kernel, general addressing)
]

# ---- fp8 weight gradients (x8: e4m3, dz8: e5m2; 128-pixel slices through ds_read_b64_tr_b8): both addressing modes, the pixel
# tail of a slice (M % 128 != 0, M < 128), the wide tile
WGRAD_FP8 = [
    dict(id="f8_c4_3x3", n=3, h=24, w=78, cin=256, cout=256, k=3, s=1, p=1),
    dict(id="f8_c4_1x1_1024_256", n=2, h=24, w=78, cin=1024, cout=256, k=1, s=1, p=0),
    dict(id="f8_small_s2", n=1, h=15, w=21, cin=256, cout=128, k=1, s=2, p=0),
    dict(id="f8_small_3x3", n=2, h=12, w=10, cin=64, cout=64, k=3, s=1, p=1),
    dict(id="f8_fpn_p2_3x3_wide", n=1, h=94, w=311, cin=256, cout=256, k=3, s=1, p=1),
]
# grouped: bf16 and fp8 layers in one table (up to four launches: {bf16, fp8} x {linear, general})
WGRAD_GROUPS_FP8 = [
    dict(id="f8_mixed", layers=[dict(n=2, h=24, w=39, cin=256, cout=256, k=3, s=1, p=1, f8=True), dict(n=2, h=24, w=39, cin=256, cout=1024, k=1, s=1, p=0, f8=True),
                                dict(n=2, h=24, w=39, cin=1024, cout=256, k=1, s=1, p=0, f8=False), dict(n=1, h=47, w=77, cin=512, cout=256, k=1, s=2, p=0, f8=True),
                                dict(n=2, h=12, w=10, cin=128, cout=128, k=3, s=1, p=1, f8=False)]),
    dict(id="f8_narrow", layers=[dict(n=2, h=12, w=10, cin=64, cout=64, k=3, s=1, p=1, f8=True), dict(n=2, h=9, w=13, cin=128, cout=256, k=1, s=1, p=0, f8=True),
                                 dict(n=2, h=10, w=12, cin=256, cout=64, k=1, s=1, p=0, f8=True)]),                       # linear, 64 output channels: 64 x 64 tiles
]

# ---- grouped weight gradients: one launch per addressing mode; tile 128 x 64 when every layer of the mode has cout >= 128
WGRAD_GROUPS = [
    dict(id="mixed_64x64", layers=[dict(n=2, h=12, w=10, cin=64, cout=64, k=3, s=1, p=1), dict(n=1, h=15, w=21, cin=256, cout=128, k=1, s=2, p=0),
                                   dict(n=2, h=9, w=13, cin=128, cout=256, k=1, s=1, p=0), dict(n=3, h=24, w=26, cin=256, cout=256, k=3, s=1, p=1),
                                   dict(n=2, h=10, w=12, cin=64, cout=64, k=1, s=1, p=0)]),
    dict(id="wide_128x64", layers=[dict(n=2, h=24, w=39, cin=256, cout=256, k=3, s=1, p=1), dict(n=2, h=24, w=39, cin=256, cout=1024, k=1, s=1, p=0),
                                   dict(n=2, h=24, w=39, cin=1024, cout=256, k=1, s=1, p=0), dict(n=1, h=47, w=77, cin=512, cout=256, k=1, s=2, p=0)]),
]


def conv_desc(ops, c, **kw):
    ho, wo = (c["h"] + 2 * c["p"] - c["k"]) // c["s"] + 1, (c["w"] + 2 * c["p"] - c["k"]) // c["s"] + 1
    return ops.conv_desc(c["n"], c["h"], c["w"], c["cin"], c["k"], c["k"], c["s"], c["p"], c["p"], ho, wo, c["cout"], **kw)


def _with_workspace(ops, c, d, device):
    if c.get("ws"):
        assert ops.conv_attach_workspace(d, device) is not None, "case %s: the dispatcher does not take a workspace here" % c["id"]
    return d


def fprop_desc(ops, c, device="cpu"):
    flags = (ops.CONV_BIAS if c["bias"] else 0) | (ops.CONV_RELU if c["relu"] else 0) | (ops.CONV_STATS if c["stats"] else 0)
    return _with_workspace(ops, c, conv_desc(ops, c, flags=flags), device)


def dgrad_desc(ops, c, device="cpu"):
    n, h, w, k, sc = c["n"], c["h"], c["w"], c["k"], c["scatter"]
    flags = ops.CONV_ADD_RES if c["res"] else 0
    if sc == 1:
        d = ops.conv_desc(n, h, w, c["cin"], k, k, 1, k // 2, k // 2, h, w, c["cout"], flags=flags)
    else:
        d = ops.conv_desc(n, h, w, c["cin"], 1, 1, 1, 0, 0, h, w, c["cout"], out_h=sc * h, out_w=sc * w - 1, out_scatter=sc, flags=flags)   # (odd width, as 311 -> 156)
    return _with_workspace(ops, c, d, device)


def f32_desc(ops, c):
    if c["split"] > 1:
        return ops.conv_desc(1, 1, c["m"], c["k"], 1, 1, 1, 0, 0, 1, c["m"], c["cout"], flags=ops.CONV_SPLITK_ATOMIC, split_k=c["split"])
    return ops.conv_desc(1, 1, c["m"], c["k"], 1, 1, 1, 0, 0, 1, c["m"], c["cout"], flags=ops.CONV_OUT_F32 | (ops.CONV_BIAS if c["bias"] else 0))


def _strip(name):
    return name.split(" grid")[0]


# ---- convolutions that carry a neighbouring BatchNorm (ops entry point -> cases).  Each case is run by an "equals its parts" GPU test
# (tests/test_gpu_kernels.py) and counted by tests/test_conv_dispatch.py under "<instantiation>+<entry point>": a plan that sends another
# instantiation through one of these entry points names the missing case.
BN_FUSED_ENTRY_POINTS = ("conv2d_fprop_bnin",)
BNIN = [   # frcnn_conv2d_fprop_bnin: 3x3 / stride 1 / pad 1 on the RAW output of the 1x1 layer before it (n, h, w = its grid)
    dict(id="bnin_c2_3x3_64_64_stats_b4", n=4, h=94, w=311, cin=64, cout=64, stats=True),              # conv2's shape at the benchmark's batch
    dict(id="bnin_3x3_64_128_nostats_ragged", n=9, h=45, w=70, cin=64, cout=128, stats=False),         # ends inside tiles in both directions, two channel parts
    dict(id="bnin_3x3_64_64_stats_ragged", n=18, h=45, w=70, cin=64, cout=64, stats=True),
    dict(id="bnin_c2_3x3_64_64_stats_600x1987_b2", n=2, h=150, w=497, cin=64, cout=64, stats=True),    # the reference's own configuration (config.json:3, batch 2)
    # round 5: the patch-resident kernel's loader-wave forms (conv3x3_patch<..., BNIN=1>): conv4 (four chunks, four channel parts: every
    # workgroup writes ONE chunk of its tile's activation), conv3 (128-channel parts: two chunks, one part), ResNet-101's batch, the
    # reference's own 600 x 1987 (odd grids), and ragged grids with an odd chunk count / fewer chunks than channel parts
    dict(id="bnin_c4_3x3_256_256_stats_b4", n=4, h=24, w=78, cin=256, cout=256, stats=True),
    dict(id="bnin_c4_3x3_256_256_stats_b2", n=2, h=24, w=78, cin=256, cout=256, stats=True),
    dict(id="bnin_c3_3x3_128_128_stats_b4", n=4, h=47, w=156, cin=128, cout=128, stats=True),
    dict(id="bnin_c3_3x3_128_128_stats_b2", n=2, h=47, w=156, cin=128, cout=128, stats=True),
    dict(id="bnin_c4_3x3_256_256_stats_600x1987_b2", n=2, h=38, w=125, cin=256, cout=256, stats=True),
    dict(id="bnin_c3_3x3_128_128_stats_600x1987_b2", n=2, h=75, w=249, cin=128, cout=128, stats=True),
    dict(id="bnin_patch_3x3_320_64_nostats_ragged", n=1, h=9, w=33, cin=320, cout=64, stats=False),      # five chunks, one part
    dict(id="bnin_patch_3x3_128_256_stats_ragged", n=4, h=13, w=21, cin=128, cout=256, stats=True),      # two chunks, four parts: parts 2 / 3 write nothing
    # round 5: 1x1 / stride-1 layers on the tile kernel (conv_tile<..., BNIN=1>: every landed A slice transformed in place) -- the third
    # convolution of a bottleneck block at the benchmark's sizes (tile runs of 8 / 4 with one / two slices; 128 x 128 tiles with four),
    # ResNet-101's batch (the three-workgroup form's shapes run the two-workgroup BNIN twin), the reference's 600 x 1987, tails
    dict(id="bnin_c2_1x1_64_256_stats_b4", n=4, h=94, w=311, cin=64, cout=256, stats=True, k=1),
    dict(id="bnin_c3_1x1_128_512_stats_b4", n=4, h=47, w=156, cin=128, cout=512, stats=True, k=1),
    dict(id="bnin_c4_1x1_256_1024_stats_b4", n=4, h=24, w=78, cin=256, cout=1024, stats=True, k=1),
    dict(id="bnin_c4_1x1_256_1024_stats_b2", n=2, h=24, w=78, cin=256, cout=1024, stats=True, k=1),
    dict(id="bnin_c2_1x1_64_256_stats_600x1987_b2", n=2, h=150, w=497, cin=64, cout=256, stats=True, k=1),
    dict(id="bnin_c3_1x1_128_512_stats_600x1987_b2", n=2, h=75, w=249, cin=128, cout=512, stats=True, k=1),
    dict(id="bnin_c4_1x1_256_1024_stats_600x1987_b2", n=2, h=38, w=125, cin=256, cout=1024, stats=True, k=1),
    dict(id="bnin_1x1_64_256_stats_tail", n=2, h=13, w=17, cin=64, cout=256, stats=True, k=1),           # M = 442: four tiles, the last with 58 rows
    dict(id="bnin_1x1_192_64_nostats_tail", n=3, h=9, w=11, cin=192, cout=64, stats=False, k=1),         # three slices, one channel part
    dict(id="bnin_1x1_128_384_relu_nostats", n=1, h=40, w=33, cin=128, cout=384, stats=False, k=1),      # six parts (not a power of two)
]


def bnin_desc(ops, c):
    k = c.get("k", 3)
    return ops.conv_desc(c["n"], c["h"], c["w"], c["cin"], k, k, 1, k // 2, k // 2, c["h"], c["w"], c["cout"], flags=ops.CONV_BIAS | (ops.CONV_STATS if c["stats"] else 0))


def covered_instantiations(ops):
    """{instantiation name: [case ids]} for every case above (dry-run of the dispatcher: no GPU needed)."""
    import torch
    out = {}

    def note(name, cid):
        out.setdefault(_strip(name), []).append(cid)

    for c in FPROP:
        note(ops.conv2d_describe(fprop_desc(ops, c), False), c["id"])
    for c in BNIN:
        note(ops.conv2d_describe(bnin_desc(ops, c), False).replace(" grid", "+conv2d_fprop_bnin grid", 1), c["id"])
    for c in FPROP_FP8:
        note(ops.conv2d_describe_fp8(fprop_desc(ops, c)), c["id"])
    for c in DGRAD_FP8:
        note(ops.conv2d_describe_dgrad_fp8(dgrad_desc(ops, c), c["red"]), c["id"])
    for c in DGRAD:
        note(ops.conv2d_describe(dgrad_desc(ops, c), c["red"]), c["id"])
    for c in F32:
        note(ops.conv2d_describe(f32_desc(ops, c), False), c["id"])
    for c in WGRAD:
        note(ops.conv2d_wgrad_describe(conv_desc(ops, c)), c["id"])
    note(ops.conv2d_wgrad_describe(ops.conv_desc(1, 1, 24, 1024, 1, 1, 1, 0, 0, 1, 24, 64), with_row_index=True), "head wgrad (row_index)")
    note(ops.conv2d_wgrad_describe(ops.conv_desc(2, 43, 52, 32, 7, 1, 2, 0, 0, 19, 23, 64, in_pix_stride=4)), "stem wgrad")
    note(ops.conv2d_describe(ops.conv_desc(2, 43, 52, 32, 7, 1, 2, 0, 0, 19, 23, 64, in_pix_stride=4, flags=ops.CONV_BIAS | ops.CONV_STATS)), "stem fprop")
    for c in WGRAD_FP8:
        note(ops.conv2d_wgrad_describe_fp8(conv_desc(ops, c)), c["id"])
    for grp in WGRAD_GROUPS + WGRAD_GROUPS_FP8:
        items = []
        for c in grp["layers"]:
            d = conv_desc(ops, c)
            it = (d, torch.zeros(8, dtype=torch.bfloat16), torch.zeros(8, dtype=torch.bfloat16), torch.zeros(8))
            items.append(it + (torch.zeros(1), torch.zeros(1)) if c.get("f8") else it)
        g = ops.WgradGroup(items, "cpu")
        for part in ops.conv2d_wgrad_describe(group=g).split("; "):
            if part.strip():
                note(part, grp["id"])
    return out
