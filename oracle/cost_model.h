// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Frozen algorithmic cost table.  bench.py applies these weights to the work
// counters to get the ALGORITHMIC bytes and flops per ray that `roofline.achieved` uses.
// Canonical f32 layout: node = min3+max3+left+right = 32 B; one primitive test = 16 B;
// one 32 B material/texture record per accepted closest hit; 12 B accumulator write per
// pixel per launch.  Flop weights count add/mul/div/sqrt/min/max as 1 each, derived from
// aabb.rs:18-31, hittable.rs:258-278, :213-253, alias.rs:112-146, material.rs:57-116,
// pdf.rs:13-39.  tests/test_cost_model.py checks bench.py's copy against this file.
#pragma once
#define RT_COST_BYTES_PER_NODE 32
#define RT_COST_BYTES_PER_PRIM_TEST 16
#define RT_COST_BYTES_PER_CLOSEST_HIT 32
#define RT_COST_BYTES_PER_PIXEL_WRITE 12
#define RT_COST_FLOPS_PER_NODE 18
#define RT_COST_FLOPS_PER_SPHERE_TEST 25
#define RT_COST_FLOPS_PER_SPHERE_HIT 17
#define RT_COST_FLOPS_PER_CUBE_TEST 18
#define RT_COST_FLOPS_PER_CUBE_HIT 30
#define RT_COST_FLOPS_PER_ROTATE_VISIT 112
#define RT_COST_FLOPS_PER_TRANSLATE_VISIT 6
#define RT_COST_FLOPS_PER_DIFFUSE_VERTEX 60
#define RT_COST_FLOPS_PER_SPECULAR_VERTEX 40
