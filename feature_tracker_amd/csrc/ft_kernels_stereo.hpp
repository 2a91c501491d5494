// Stereo matching geometry around the LK launch, for gfx950: what TrackMsckfVio::stereoMatch does before and after
// its calcOpticalFlowPyrLK call (reference: src/track/TrackMsckfVio.cpp:218-301).
//
//   stereo_seed_kernel   initial guess of every cam0 point in cam1 under the extrinsic rotation alone:
//                        cv::undistortPoints(K0, D0, R_cam0_cam1) -> Point2f -> cv::projectPoints(K1, D1) (:238-249)
//   stereo_check_kernel  after tracking: points that left the image are dropped (:259-268), then points further than
//                        stereo_threshold pixels from their epipolar line under E = [t]x R (:270-298).  The kernel
//                        can also move the batch results into the pinned host block.
//
// Double precision in a fixed order (results equal the CPU checker's bit for bit); one thread per point.
#pragma once
#include "ft_kernels_outlier.hpp"

namespace ftd {

struct StereoArgs {
    const LKJobDev* jobs;
    const uint16_t* pt_job;    // job index of every point
    const float* prev_xy;      // device, cam0 points
    float* next_xy;            // device: guesses (seed kernel writes) / LK results (check kernel reads)
    const float* err;          // device
    uint8_t* status;           // device, updated in place by the check
    float* out_xy;             // pinned host, written when emit != 0
    float* out_err;
    uint8_t* out_status;
    int total;
    int emit;
    double threshold;          // pixels; <= 0: no epipolar test
};

__global__ void __launch_bounds__(256) stereo_seed_kernel(StereoArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const LKJobDev* job = a.jobs + a.pt_job[i];
    if (!job->use_stereo) return;
    const double* c0 = job->calib;
    const double* c1 = job->calib + 8;
    const double* R = job->stereo;
    double x, y;
    ro_undistort_d(a.prev_xy[2 * i], a.prev_xy[2 * i + 1], c0, x, y);
    const double X = R[0] * x + R[1] * y + R[2], Y = R[3] * x + R[4] * y + R[5], W = R[6] * x + R[7] * y + R[8];
    const double iw = 1. / W;
    const double xr = (double)(float)(X * iw), yr = (double)(float)(Y * iw);   // cv::Point2f between the two calls
    const double r2 = xr * xr + yr * yr, r4 = r2 * r2;
    const double cdist = 1. + c1[4] * r2 + c1[5] * r4;
    const double a1 = 2. * xr * yr, a2 = r2 + 2. * xr * xr, a3 = r2 + 2. * yr * yr;
    const double xd = xr * cdist + c1[6] * a1 + c1[7] * a2;
    const double yd = yr * cdist + c1[6] * a3 + c1[7] * a1;
    a.next_xy[2 * i] = (float)(xd * c1[0] + c1[2]);
    a.next_xy[2 * i + 1] = (float)(yd * c1[1] + c1[3]);
}

__global__ void __launch_bounds__(256) stereo_check_kernel(StereoArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const LKJobDev* job = a.jobs + a.pt_job[i];
    const float nx = a.next_xy[2 * i], ny = a.next_xy[2 * i + 1];
    uint8_t st = a.status[i];
    if (job->use_stereo && st) {
        if (ny < 0.f || ny > (float)(job->h0 - 1) || nx < 0.f || nx > (float)(job->w0 - 1)) st = 0;
        if (st && a.threshold > 0.) {
            const double* c0 = job->calib;
            const double* c1 = job->calib + 8;
            const double* R = job->stereo;
            const double* t = job->stereo + 9;
            const float2 p0 = ro_undistort(a.prev_xy[2 * i], a.prev_xy[2 * i + 1], c0);
            const float2 p1 = ro_undistort(nx, ny, c1);
            // E = [t]x R, row by row
            const double E0 = -t[2] * R[3] + t[1] * R[6], E1 = -t[2] * R[4] + t[1] * R[7], E2 = -t[2] * R[5] + t[1] * R[8];
            const double E3 = t[2] * R[0] - t[0] * R[6], E4 = t[2] * R[1] - t[0] * R[7], E5 = t[2] * R[2] - t[0] * R[8];
            const double E6 = -t[1] * R[0] + t[0] * R[3], E7 = -t[1] * R[1] + t[0] * R[4], E8 = -t[1] * R[2] + t[0] * R[5];
            const double x0 = p0.x, y0 = p0.y, x1 = p1.x, y1 = p1.y;
            const double l0 = E0 * x0 + E1 * y0 + E2, l1 = E3 * x0 + E4 * y0 + E5, l2 = E6 * x0 + E7 * y0 + E8;
            const double error = fabs(x1 * l0 + y1 * l1 + l2) / sqrt(l0 * l0 + l1 * l1);
            const double unit = 4.0 / (c0[0] + c0[1] + c1[0] + c1[1]);
            if (error > a.threshold * unit) st = 0;
        }
        a.status[i] = st;
    }
    if (a.emit) {
        a.out_xy[2 * i] = nx;
        a.out_xy[2 * i + 1] = ny;
        a.out_err[i] = a.err[i];
        a.out_status[i] = st;
    }
}

}  // namespace ftd
