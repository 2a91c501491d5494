// Visual tracker base class (call surface of src/track/TrackBase.h:72-368) without OpenCV/Eigen types.
// Images are handed in as borrowed Image8 views and copied to the device on entry, so the caller may reuse its
// buffer as soon as feed_* returns (the reference clones the cv::Mat, src/track/TrackKLT.cpp:46-47).
// The drawing helpers display_active/display_history are GUI code and are not part of this build; the data they
// draw is available through get_last_points()/get_last_ids().
#pragma once
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../feat/FeatureDatabase.h"
#include "../propagator/TrackPropagator.h"
#include "../types.h"

namespace feature_tracker {

class TrackBase {
public:
    TrackBase() : database(new FeatureDatabase()), propagator(new TrackPropagator()), num_features(200), currid(0) {}
    // numaruco: ids start above the ArUco tag range (TrackBase.h:86-89)
    TrackBase(int numfeats, int numaruco)
        : database(new FeatureDatabase()), propagator(new TrackPropagator()), num_features(numfeats), currid((size_t)numaruco + 1) {
        // thousands of observations per frame: the database is split by feature id so that several host threads can
        // write one frame's observations side by side
        if (numfeats >= 1024) database->set_shards(4);
    }
    virtual ~TrackBase() = default;

    // camera_calib: fx,fy,cx,cy,d1..d4 per camera; camera_extrinsics: 4x4 T_cam_imu per camera (TrackBase.h:99-215)
    void set_calibration(const std::map<size_t, Calib8>& camera_calib, const std::map<size_t, Mat4d>& camera_extrinsics,
                         bool correct_active = false) {
        const bool first_time = mtx_feeds.empty() || camera_k.empty();
        if (first_time) {
            std::vector<std::mutex> list(camera_calib.size());
            mtx_feeds.swap(list);
        }
        for (const auto& cam : camera_calib) {
            std::unique_ptr<std::lock_guard<std::mutex>> lck;
            if (!first_time && cam.first < mtx_feeds.size()) lck.reset(new std::lock_guard<std::mutex>(mtx_feeds[cam.first]));
            camera_k[cam.first] = {{cam.second[0], cam.second[1], cam.second[2], cam.second[3]}};
            camera_d[cam.first] = {{cam.second[4], cam.second[5], cam.second[6], cam.second[7]}};
        }
        if (first_time) {
            for (const auto& cam : camera_extrinsics) {
                // R_cam_imu = R_imu_cam^T, t_cam_imu = -R_imu_cam^T t_imu_cam (TrackBase.h:135-150)
                Mat3d Rt;
                Vec3d t;
                for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rt[3 * r + c] = cam.second[4 * c + r];
                for (int r = 0; r < 3; r++)
                    t[r] = -(Rt[3 * r] * cam.second[3] + Rt[3 * r + 1] * cam.second[7] + Rt[3 * r + 2] * cam.second[11]);
                R_cam_imu[cam.first] = Rt;
                t_cam_imu[cam.first] = t;
            }
            propagator->set_imu_cam_calib(camera_extrinsics);
            return;
        }
        if (correct_active) {   // re-normalize everything in the database with the new intrinsics
            for (const auto& kv : database->get_internal_data()) {
                Feature* feat = kv.second;
                for (const auto& meas : feat->timestamps) {
                    const size_t camid = meas.first;
                    for (size_t m = 0; m < feat->uvs.at(camid).size(); m++) {
                        const Point2f n = undistort_point(Point2f(feat->uvs.at(camid)[m][0], feat->uvs.at(camid)[m][1]), camid);
                        feat->uvs_norm.at(camid)[m] = {{n.x, n.y}};
                    }
                }
            }
        }
    }

    // Returns FT_OK or the first error code of the device layer (the reference's feed_stereo returns void,
    // src/track/TrackBase.h:225: there the OpenCV calls throw instead).
    virtual int feed_stereo(double timestamp, const Image8& img_left, const Image8& img_right, size_t cam_id_left, size_t cam_id_right) = 0;

    FeatureDatabase* get_feature_database() { return database.get(); }

    // rename an actively tracked feature (TrackBase.h:257-276)
    void change_feat_id(size_t id_old, size_t id_new) {
        database->rekey(id_old, id_new);
        for (auto& cam_ids : ids_last)
            for (auto& id : cam_ids.second)
                if (id == id_old) id = id_new;
    }

    // raw pixel -> undistorted normalized coordinates, radtan/brown model (TrackBase.h:289-332)
    Point2f undistort_point(Point2f pt_in, size_t cam_id) const {
        return undistort_point_brown(pt_in, camera_k.at(cam_id), camera_d.at(cam_id));
    }

    // (timestamp, accel, gyro), as TrackBase.h:303-308
    void feed_imu(double timestamp, const Vec3d& am, const Vec3d& wm) { propagator->push_imu(timestamp, wm, am); }

    std::vector<KeyPoint> get_last_points(size_t cam_id) {
        auto it = pts_last.find(cam_id);
        return it == pts_last.end() ? std::vector<KeyPoint>() : it->second;
    }
    std::vector<size_t> get_last_ids(size_t cam_id) {
        auto it = ids_last.find(cam_id);
        return it == ids_last.end() ? std::vector<size_t>() : it->second;
    }
    int get_num_features() const { return num_features; }

protected:
    // what cv::undistortPoints does for one point: remove K, then five fixed-point iterations of the inverse
    // radial-tangential model
    static Point2f undistort_point_brown(Point2f pt_in, const std::array<double, 4>& K, const std::array<double, 4>& D) {
        // cv::undistortPoints multiplies by the reciprocal focal lengths (ifx = 1./fx)
        const double ifx = 1. / K[0], ify = 1. / K[1];
        const double x0 = ((double)pt_in.x - K[2]) * ifx, y0 = ((double)pt_in.y - K[3]) * ify;
        double x = x0, y = y0;
        if (D[0] != 0. || D[1] != 0. || D[2] != 0. || D[3] != 0.) {
            for (int it = 0; it < 5; it++) {
                const double r2 = x * x + y * y;
                const double icdist = 1. / (1. + (D[1] * r2 + D[0]) * r2);
                const double dx = 2. * D[2] * x * y + D[3] * (r2 + 2. * x * x);
                const double dy = D[2] * (r2 + 2. * y * y) + 2. * D[3] * x * y;
                x = (x0 - dx) * icdist;
                y = (y0 - dy) * icdist;
            }
        }
        return Point2f((float)x, (float)y);
    }

    std::unique_ptr<FeatureDatabase> database;
    std::unique_ptr<TrackPropagator> propagator;
    std::map<size_t, std::array<double, 4>> camera_k;   // fx, fy, cx, cy
    std::map<size_t, std::array<double, 4>> camera_d;   // k1, k2, p1, p2
    std::map<size_t, Mat3d> R_cam_imu;
    std::map<size_t, Vec3d> t_cam_imu;
    int num_features;
    std::vector<std::mutex> mtx_feeds;
    std::unordered_map<size_t, std::vector<KeyPoint>> pts_last;
    std::unordered_map<size_t, std::vector<size_t>> ids_last;
    std::atomic<size_t> currid;
};

}  // namespace feature_tracker
