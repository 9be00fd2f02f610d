// hg_match_big.h — entry points of the -DHG_BIG compilation of hg_match.hip (the same functions on problems
// of up to 48 control points / 160 TSDF blocks / 96 odometry and IMU blocks), as the plain build calls them.
// Not part of the public ABI (include/hg_mi355x.h): a problem is promoted behind hg_problem_*.
#pragma once
#include "../../include/hg_mi355x.h"

struct hg_problem_big;
extern "C" {
int hg_problem_create_big(hg_ctx* ctx, hg_problem_big** out);
int hg_problem_destroy_big(hg_problem_big* p);
int hg_problem_reset_big(hg_problem_big* p);
int hg_problem_add_pose_big(hg_problem_big* p, const double tq[7], int constant);
int hg_problem_set_pose_big(hg_problem_big* p, int index, const double tq[7]);
int hg_problem_get_pose_big(hg_problem_big* p, int index, double tq[7]);
int hg_problem_set_velocity_big(hg_problem_big* p, int index, const double v[3], int constant);
int hg_problem_get_velocity_big(hg_problem_big* p, int index, double v[3]);
int hg_problem_add_odometry_block_big(hg_problem_big* p, int pose_a, int pose_b, double translation_weight,
                                      double rotation_weight, const double delta_tq[7]);
int hg_problem_add_imu_block_big(hg_problem_big* p, int pose_a, int pose_b, double translation_weight,
                                 double velocity_weight, double rotation_weight, double delta_time_seconds,
                                 const double delta_rotation_wxyz[4]);
int hg_problem_add_block_big(hg_problem_big* p, const float* xyz, size_t n, int memspace, hg_grid* const* pyramid,
                             int levels, int multi_res, double scaling_factor, int pose_a, int pose_b,
                             double interpolation_ratio);
int hg_problem_add_unwarped_block_big(hg_problem_big* p, const float* xyz, const double* interpolation_ratios,
                                      size_t n, int memspace, hg_grid* const* pyramid, int levels, int multi_res,
                                      double scaling_factor, int pose_a, int pose_b);
int hg_problem_set_block_width_big(hg_problem_big* p, int block, size_t width);
int hg_problem_evaluate_big(hg_problem_big* p, double* cost, double* residuals, double* gradient, double* JtJ);
int hg_problem_solve_async_big(hg_problem_big* p, const hg_solver_opts* opts);
int hg_problem_fetch_big(hg_problem_big* p, hg_solver_summary* summary);
}
namespace hg {
// Device address of the solved control poses of a big problem (pose k at base + k * stride doubles).
const double* big_device_poses(hg_problem_big* p, int* stride);
void big_orphan(hg_problem_big* p);
}
