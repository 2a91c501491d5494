// IMU buffer and mean-gyro rotation between two camera frames (call surface of
// src/propagator/TrackPropagator.h:22-213).  The reference never calls integrate_by_msckfVio and its window
// end test reads the wrong iterator (:122); this restatement implements the intended msckf_vio behaviour:
// average the gyro over [last - 0.01 s, cur + 0.005 s), rotate it into each camera frame, R_p_c = exp(w dt)^T.
#pragma once
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "../types.h"

namespace feature_tracker {

class TrackPropagator {
public:
    struct IMUDATA {
        double timestamp;
        Vec3d wm;   // gyroscope, rad/s
        Vec3d am;   // accelerometer, m/s^2
    };

    // (timestamp, gyro, accel).  The reference declares the parameters as (timestamp, am, wm) but is called with
    // (timestamp, wm, am) from TrackBase::feed_imu (TrackBase.h:307), so the gyro ends up being averaged; here the
    // names say what the values are.
    void push_imu(double timestamp, const Vec3d& wm, const Vec3d& am) {
        std::lock_guard<std::mutex> lck(mtx_imu);
        imu_data.push_back(IMUDATA{timestamp, wm, am});
    }

    // rotation blocks of T_cam_imu for cameras 0 and 1 (TrackPropagator.h:81-102)
    void set_imu_cam_calib(const std::map<size_t, Mat4d>& T_imu_cam) {
        std::lock_guard<std::mutex> lck(mtx_imu);
        for (const auto& kv : T_imu_cam) {
            Mat3d R;
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = kv.second[4 * r + c];
            R_imu_to_cam[kv.first] = R;
        }
    }

    // mean-gyro relative rotation previous->current for cam0/cam1 (TrackPropagator.h:107-153)
    int integrate_by_msckfVio(double lastTime, double currTime, Mat3f& cam0_R_p_c, Mat3f& cam1_R_p_c, bool remove = false) {
        std::lock_guard<std::mutex> lck(mtx_imu);
        size_t b = head;
        while (b < imu_data.size() && (imu_data[b].timestamp - lastTime) < -0.01) b++;
        size_t e = b;
        while (e < imu_data.size() && (imu_data[e].timestamp - currTime) < 0.005) e++;
        float mean[3] = {0.f, 0.f, 0.f};
        for (size_t k = b; k < e; k++) for (int a = 0; a < 3; a++) mean[a] += (float)imu_data[k].wm[a];
        if (e > b) { const float s = 1.0f / (float)(e - b); for (int a = 0; a < 3; a++) mean[a] *= s; }
        const double dt = currTime - lastTime;
        rotation_for(0, mean, dt, cam0_R_p_c);
        rotation_for(1, mean, dt, cam1_R_p_c);
        const int used = (int)(e - b);
        // samples before the window are spent: move the head past them, and compact the buffer only now and then
        head = remove ? e : b;
        if (head > 256 && head > imu_data.size() / 2) {
            imu_data.erase(imu_data.begin(), imu_data.begin() + head);
            head = 0;
        }
        return used;
    }

    size_t buffered() {
        std::lock_guard<std::mutex> lck(mtx_imu);
        return imu_data.size() - head;
    }

protected:
    void rotation_for(size_t cam, const float* mean, double dt, Mat3f& R_p_c) const {
        Mat3d Ric{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
        auto it = R_imu_to_cam.find(cam);
        if (it != R_imu_to_cam.end()) Ric = it->second;
        float wc[3];
        for (int r = 0; r < 3; r++)
            wc[r] = (float)Ric[3 * r] * mean[0] + (float)Ric[3 * r + 1] * mean[1] + (float)Ric[3 * r + 2] * mean[2];
        const double rx = (double)(float)((double)wc[0] * dt), ry = (double)(float)((double)wc[1] * dt), rz = (double)(float)((double)wc[2] * dt);
        const double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (theta >= 2.220446049250313e-16) {   // Rodrigues
            const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c;
            const double kx = rx / theta, ky = ry / theta, kz = rz / theta;
            R[0] = c + c1 * kx * kx;      R[1] = c1 * kx * ky - s * kz; R[2] = c1 * kx * kz + s * ky;
            R[3] = c1 * kx * ky + s * kz; R[4] = c + c1 * ky * ky;      R[5] = c1 * ky * kz - s * kx;
            R[6] = c1 * kx * kz - s * ky; R[7] = c1 * ky * kz + s * kx; R[8] = c + c1 * kz * kz;
        }
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) R_p_c[3 * r + cc] = (float)R[3 * cc + r];   // transposed
    }

    std::mutex mtx_imu;
    std::vector<IMUDATA> imu_data;
    size_t head = 0;   // first sample not yet spent
    std::map<size_t, Mat3d> R_imu_to_cam;
};

}  // namespace feature_tracker
