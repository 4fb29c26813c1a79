// Debug aid of tests/tools/bisect_sim_o3.py (not part of the product build): PARC_LOOP(k) keeps the loops tagged k rolled when
// PARC_ROLL_<k> is defined.  Tags: 0 kinematics sweep, 1 rigid-body inertia, 2 contact impedance, 3 contact report, 4 inward sweep,
// 5 spherical drive (3 axes), 6 outward acceleration sweep, 7 integration, 8 load_state, 9 store_state, 10 publish_bodies,
// 11 substep loop, 12 the fixed-trip 3x3 helpers, 13 neighbour columns of sphere_vs_columns.
#pragma once
#define PARC_LOOP(k) PARC_LOOP_##k
#ifdef PARC_ROLL_0
#define PARC_LOOP_0 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_0
#endif
#ifdef PARC_ROLL_1
#define PARC_LOOP_1 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_1
#endif
#ifdef PARC_ROLL_2
#define PARC_LOOP_2 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_2
#endif
#ifdef PARC_ROLL_3
#define PARC_LOOP_3 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_3
#endif
#ifdef PARC_ROLL_4
#define PARC_LOOP_4 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_4
#endif
#ifdef PARC_ROLL_5
#define PARC_LOOP_5 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_5
#endif
#ifdef PARC_ROLL_6
#define PARC_LOOP_6 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_6
#endif
#ifdef PARC_ROLL_7
#define PARC_LOOP_7 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_7
#endif
#ifdef PARC_ROLL_8
#define PARC_LOOP_8 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_8
#endif
#ifdef PARC_ROLL_9
#define PARC_LOOP_9 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_9
#endif
#ifdef PARC_ROLL_10
#define PARC_LOOP_10 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_10
#endif
#ifdef PARC_ROLL_11
#define PARC_LOOP_11 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_11
#endif
#ifdef PARC_ROLL_12
#define PARC_LOOP_12 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_12
#endif
#ifdef PARC_ROLL_13
#define PARC_LOOP_13 _Pragma("clang loop unroll(disable)")
#else
#define PARC_LOOP_13
#endif
