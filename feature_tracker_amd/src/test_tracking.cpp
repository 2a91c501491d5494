// Driver of the stereo KLT front-end with the flow of the reference's driver (src/test_tracking.cpp:220-246):
//   load_params -> new TrackKLT(num_pts, 0, fast_threshold, grid_x, grid_y, min_px_dist) -> set_calibration
//   -> feed every IMU sample (a thread that is joined at once, :234-235) -> feed_stereo per image pair (:191-216).
//
// What differs, and why:
//  * the dataset folder and the calibration file are arguments, not absolute paths of the author's machine (:24,:121,:156);
//  * images are binary PGM (left/<t>.pgm, right/<t>.pgm): there is no JPEG decoder in this build (the reference
//    reads left/<t>.jpg through cv::imread, :178-199);
//  * timestamps are handed over in seconds with their fraction.  The reference truncates them to whole seconds
//    (`(uint64_t)(vTimeStamp[ni]*1e-9)`, :207), which would give a zero interval to the gyro seed;
//  * the time window 40 s .. 325 s of the author's recording (:193) is an option (--t-min / --t-max, seconds);
//  * no window is opened (display_active / imshow, :210-213): per-frame track counts are printed, and --dump writes
//    every camera's ids and positions per frame to a text file;
//  * --write-dataset makes a synthetic recording in the same folder layout (a textured plane seen by a rig with a
//    constant disparity, in-plane roll with matching gyro samples), so the driver can be run without a recording.
//
// The backend comes from ftk_make_backend(): the product binary links the gfx950 device library and fails without
// a GPU; the test suite links the same file against its CPU checker.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

#include "FeatureTrackerOptions.h"
#include "track/TrackKLT.h"
#include "utils/calib_yaml.h"

using namespace feature_tracker;

FrontendBackend* ftk_make_backend(int device, const ft_limits& limits, void** device_ctx);
const char* ftk_backend_label();

extern "C" {
void fts_canvas(uint8_t* out, int cw, int ch, uint32_t seed, int n_waves, int n_shapes, int threads);
void fts_frame(const uint8_t* canvas, int cw, int ch, uint8_t* out, int w, int h, size_t ostride, double ccx, double ccy, double theta,
               double sigma, uint32_t noise_seed, int threads);
}

namespace {

/// Our sparse feature tracker
TrackBase* trackFEATS = nullptr;

std::string data_folder = "./dataset/";
std::string param_file = "./dataset/config.yaml";
double t_min = -1e300, t_max = 1e300;
std::string dump_file;

// ---- calibration: src/test_tracking.cpp:27-116 ----
bool load_params(FeatureTrackerOptions& params) {
    CalibYaml fs;
    if (!fs.open(param_file)) {
        std::cout << "ERROR: Wrong path to settings" << std::endl;
        return false;
    }
    const YamlMatrix K_l = fs.matrix("LEFT.K"), K_r = fs.matrix("RIGHT.K");
    const YamlMatrix R_l = fs.matrix("LEFT.R"), R_r = fs.matrix("RIGHT.R");     // 4x4 T_cam_imu despite the name
    const YamlMatrix D_l = fs.matrix("LEFT.D"), D_r = fs.matrix("RIGHT.D");
    const int rows_l = (int)fs.number("LEFT.height"), cols_l = (int)fs.number("LEFT.width");
    const int rows_r = (int)fs.number("RIGHT.height"), cols_r = (int)fs.number("RIGHT.width");
    if (K_l.empty() || K_r.empty() || R_l.empty() || R_r.empty() || D_l.empty() || D_r.empty() || rows_l == 0 || rows_r == 0 ||
        cols_l == 0 || cols_r == 0 || K_l.data.size() < 9 || K_r.data.size() < 9 || R_l.data.size() < 12 || R_r.data.size() < 12 ||
        D_l.data.size() < 4 || D_r.data.size() < 4) {
        std::cout << "ERROR: Calibration parameters to rectify stereo are missing!" << std::endl;
        return false;
    }
    for (int i = 0; i < 2; i++) {
        const YamlMatrix& K = i == 0 ? K_l : K_r;
        const YamlMatrix& D = i == 0 ? D_l : D_r;
        const YamlMatrix& R = i == 0 ? R_l : R_r;
        const Calib8 cam_calib = {{K.at(0, 0), K.at(1, 1), K.at(0, 2), K.at(1, 2), D.at(0), D.at(1), D.at(2), D.at(3)}};
        Mat4d T{};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) T[4 * r + c] = R.at(r, c);
        T[15] = 1.0;
        params.camera_intrinsics[(size_t)i] = cam_calib;
        params.camera_extrinsics_mat[(size_t)i] = T;
        params.camera_wh[(size_t)i] = i == 0 ? std::make_pair(cols_l, rows_l) : std::make_pair(cols_r, rows_r);
    }
    return true;
}

// ---- IMU: src/test_tracking.cpp:119-151.  Lines of imu.txt: t[us] ax ay az gx gy gz ----
void feed_measurement_imu() {
    const std::string imuFile = data_folder + "imu.txt";
    std::ifstream f(imuFile.c_str());
    if (!f.is_open()) {
        std::cout << "imu data path is not open! file: " << imuFile << std::endl;
        return;
    }
    std::string s;
    size_t n = 0;
    while (std::getline(f, s)) {
        if (s.empty()) continue;
        std::stringstream ss(s);
        double stamp_us;
        Vec3d acc, gyr;
        if (!(ss >> stamp_us >> acc[0] >> acc[1] >> acc[2] >> gyr[0] >> gyr[1] >> gyr[2])) continue;
        trackFEATS->feed_imu(stamp_us * 1e-6, acc, gyr);
        n++;
    }
    std::cout << "IMU samples: " << n << std::endl;
}

bool read_pgm(const std::string& path, std::vector<uint8_t>& pix, int& w, int& h) {
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f.is_open()) return false;
    std::string magic;
    f >> magic;
    if (magic != "P5") return false;
    int vals[3], got = 0;
    while (got < 3 && f) {
        f >> std::ws;
        if (f.peek() == '#') { std::string c; std::getline(f, c); continue; }
        if (!(f >> vals[got])) return false;
        got++;
    }
    if (got < 3 || vals[0] < 1 || vals[1] < 1 || vals[2] != 255) return false;
    f.get();   // the single whitespace byte after maxval
    w = vals[0]; h = vals[1];
    pix.resize((size_t)w * h);
    f.read(reinterpret_cast<char*>(pix.data()), (std::streamsize)pix.size());
    return (size_t)f.gcount() == pix.size();
}

bool write_pgm(const std::string& path, const uint8_t* pix, int w, int h) {
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f.is_open()) return false;
    f << "P5\n" << w << " " << h << "\n255\n";
    f.write(reinterpret_cast<const char*>(pix), (std::streamsize)((size_t)w * h));
    return (bool)f;
}

// ---- images: src/test_tracking.cpp:154-217.  Lines of timestamp.txt: t[us]; images left/<t>.pgm, right/<t>.pgm ----
void feed_measurement_stereo() {
    const std::string strFile = data_folder + "timestamp.txt";
    std::ifstream f(strFile.c_str());
    if (!f.is_open()) {
        std::cout << "Image path is not open! file: " << strFile << std::endl;
        return;
    }
    std::vector<std::string> left_files, right_files;
    std::vector<double> stamps;
    std::string s;
    while (std::getline(f, s)) {
        if (s.empty()) continue;
        std::stringstream ss(s);
        long long t;
        if (!(ss >> t)) continue;
        stamps.push_back((double)t * 1e-6);
        left_files.push_back("left/" + std::to_string(t) + ".pgm");
        right_files.push_back("right/" + std::to_string(t) + ".pgm");
    }
    std::cout << "Image size is: " << left_files.size() << std::endl;

    std::ofstream dump;
    if (!dump_file.empty()) dump.open(dump_file.c_str());
    std::vector<uint8_t> im_left, im_right;
    std::vector<double> ms;
    for (size_t ni = 0; ni < left_files.size(); ni++) {
        if (stamps[ni] < t_min || stamps[ni] > t_max) continue;
        int wl = 0, hl = 0, wr = 0, hr = 0;
        if (!read_pgm(data_folder + left_files[ni], im_left, wl, hl) || !read_pgm(data_folder + right_files[ni], im_right, wr, hr) ||
            wl != wr || hl != hr) {
            std::cout << "Image is empty! " << data_folder + left_files[ni] << std::endl;
            continue;
        }
        const auto c0 = std::chrono::steady_clock::now();
        const int frc = trackFEATS->feed_stereo(stamps[ni], Image8(im_left.data(), wl, hl, (size_t)wl), Image8(im_right.data(), wr, hr, (size_t)wr), 0, 1);
        if (frc != FT_OK) {   // a device failure, not a lost track: stop instead of feeding on
            std::cout << "feed_stereo failed with device error " << frc << " at " << left_files[ni] << std::endl;
            break;
        }
        const auto c1 = std::chrono::steady_clock::now();
        ms.push_back(std::chrono::duration<double, std::milli>(c1 - c0).count());

        // what display_active would draw: the active tracks of both cameras
        const std::vector<size_t> idl = trackFEATS->get_last_ids(0), idr = trackFEATS->get_last_ids(1);
        const std::vector<KeyPoint> pl = trackFEATS->get_last_points(0), pr = trackFEATS->get_last_points(1);
        printf("frame %zu t=%.6f left %zu right %zu database %zu  %.3f ms\n", ni, stamps[ni], pl.size(), pr.size(),
               trackFEATS->get_feature_database()->size(), ms.back());
        if (dump.is_open()) {
            char buf[96];
            snprintf(buf, sizeof buf, "frame %zu %.6f %zu %zu\n", ni, stamps[ni], pl.size(), pr.size());
            dump << buf;
            for (int cam = 0; cam < 2; cam++) {
                const auto& ids = cam == 0 ? idl : idr;
                const auto& pts = cam == 0 ? pl : pr;
                for (size_t i = 0; i < pts.size(); i++) {
                    snprintf(buf, sizeof buf, "%d %zu %.9g %.9g\n", cam, ids[i], pts[i].pt.x, pts[i].pt.y);
                    dump << buf;
                }
            }
        }
    }
    if (!ms.empty()) {
        std::vector<double> sorted(ms.begin() + (ms.size() > 2 ? 1 : 0), ms.end());   // the first frame only detects
        std::sort(sorted.begin(), sorted.end());
        printf("feed_stereo: %zu frames, p50 %.3f ms, p95 %.3f ms\n", ms.size(), sorted[sorted.size() / 2],
               sorted[std::min(sorted.size() - 1, (size_t)(0.95 * sorted.size()))]);
    }
}

// ---- synthetic recording in the folder layout the two readers above expect ----
int write_dataset(const std::string& folder, int w, int h, int n_frames, int disparity, uint32_t seed) {
    ::mkdir(folder.c_str(), 0755);
    ::mkdir((folder + "left").c_str(), 0755);
    ::mkdir((folder + "right").c_str(), 0755);
    const int cw = 2 * w, ch = 2 * h;
    std::vector<uint8_t> canvas((size_t)cw * ch), img((size_t)w * h);
    fts_canvas(canvas.data(), cw, ch, seed, 64, std::max(50, (int)(280.0 * cw * ch / 1e6)), 4);
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const double fps = 20.0, imu_hz = 200.0, t0 = 100.0;
    const double w1 = 2 * M_PI / (55 + 20 * U(rng)), w2 = 2 * M_PI / (30 + 15 * U(rng));
    const double p1 = 2 * M_PI * U(rng), p2 = 2 * M_PI * U(rng), p3 = 2 * M_PI * U(rng), p4 = 2 * M_PI * U(rng);
    const double vbound = 0.7 * w1 + 0.3 * w2;
    const double ax = std::min(0.22 * w, 6.0 / vbound), ay = std::min(0.22 * h, 6.0 / vbound);
    const double wr = 2 * M_PI / (50 + 30 * U(rng)), amp = (0.5 * M_PI / 180.0) / wr, pr = 2 * M_PI * U(rng);
    std::ofstream ts((folder + "timestamp.txt").c_str());
    for (int k = 0; k < n_frames; k++) {
        const long long t_us = (long long)std::llround((t0 + k / fps) * 1e6);
        const double cx = cw / 2.0 + ax * (0.7 * std::sin(w1 * k + p1) + 0.3 * std::sin(w2 * k + p2));
        const double cy = ch / 2.0 + ay * (0.7 * std::cos(w1 * k + p3) + 0.3 * std::sin(w2 * k + p4));
        const double theta = amp * std::sin(wr * k + pr);
        // the right camera sees the same plane `disparity` pixels further along the (rolled) image x axis
        fts_frame(canvas.data(), cw, ch, img.data(), w, h, (size_t)w, cx, cy, theta, 2.0, seed * 1000003u + 2u * k, 4);
        if (!write_pgm(folder + "left/" + std::to_string(t_us) + ".pgm", img.data(), w, h)) return 1;
        fts_frame(canvas.data(), cw, ch, img.data(), w, h, (size_t)w, cx + disparity * std::cos(theta), cy + disparity * std::sin(theta), theta,
                  2.0, seed * 1000003u + 2u * k + 1u, 4);
        if (!write_pgm(folder + "right/" + std::to_string(t_us) + ".pgm", img.data(), w, h)) return 1;
        ts << t_us << "\n";
    }
    // gyro: the roll rate about the optical axis (camera frame = IMU frame), accel: gravity
    std::ofstream imu((folder + "imu.txt").c_str());
    imu.precision(12);
    for (double t = t0 - 0.1; t < t0 + (n_frames - 1) / fps + 0.1; t += 1.0 / imu_hz) {
        const double rate = amp * (wr * fps) * std::cos(wr * fps * (t - t0) + pr);
        imu << (long long)std::llround(t * 1e6) << " 0 0 9.81 0 0 " << rate << "\n";
    }
    // calibration in the FileStorage layout of build_pc/config/test.yaml
    const double fl = 0.61 * w, baseline = 0.11;
    std::ofstream y((folder + "config.yaml").c_str());
    y.precision(12);
    y << "%YAML:1.0\n\n";
    for (int c = 0; c < 2; c++) {
        const char* side = c == 0 ? "LEFT" : "RIGHT";
        y << side << ".height: " << h << "\n" << side << ".width: " << w << "\n";
        y << side << ".D: !!opencv-matrix\n   rows: 1\n   cols: 5\n   dt: d\n   data:[0.0, 0.0, 0.0, 0.0, 0.0]\n";
        y << side << ".K: !!opencv-matrix\n   rows: 3\n   cols: 3\n   dt: d\n   data: [" << fl << ", 0.0, " << (w - 1) / 2.0 << ", 0.0, " << fl
          << ", " << (h - 1) / 2.0 << ", 0.0, 0.0, 1.0]\n";
        y << side << ".R:  !!opencv-matrix\n   rows: 4\n   cols: 4\n   dt: d\n   data: [\n          1.0, 0.0, 0.0, " << (c == 0 ? 0.0 : baseline)
          << ",\n          0.0, 1.0, 0.0, 0.0,\n          0.0, 0.0, 1.0, 0.0,\n          0.0, 0.0, 0.0, 1.0\n         ]\n\n";
    }
    std::cout << "wrote " << n_frames << " stereo pairs of " << w << "x" << h << " to " << folder << std::endl;
    return 0;
}

void usage() {
    std::cout << "test_tracking [--data FOLDER] [--config FILE] [--dump FILE] [--t-min S] [--t-max S]\n"
                 "              [--num-pts N] [--fast-threshold N] [--grid X Y] [--min-px-dist N]\n"
                 "              [--win N] [--levels N] [--gyro-seed] [--forward-backward] [--no-reject-outliers]\n"
                 "              [--stereo-extrinsic-seed] [--device N]\n"
                 "test_tracking --write-dataset FOLDER [--size W H] [--frames N] [--disparity PX] [--seed N]\n"
                 "test_tracking --check-config FILE\n";
}

}  // namespace

int main(int argc, char** argv) {
    FeatureTrackerOptions params;
    std::string write_folder, check_file;
    int syn_w = 752, syn_h = 480, syn_frames = 30, syn_disp = 5, device = 0;
    uint32_t syn_seed = 777;
    bool config_given = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](int n) {
            if (i + n >= argc) { usage(); exit(2); }
        };
        if (a == "--data") { need(1); data_folder = argv[++i]; }
        else if (a == "--config") { need(1); param_file = argv[++i]; config_given = true; }
        else if (a == "--dump") { need(1); dump_file = argv[++i]; }
        else if (a == "--t-min") { need(1); t_min = atof(argv[++i]); }
        else if (a == "--t-max") { need(1); t_max = atof(argv[++i]); }
        else if (a == "--num-pts") { need(1); params.num_pts = atoi(argv[++i]); }
        else if (a == "--fast-threshold") { need(1); params.fast_threshold = atoi(argv[++i]); }
        else if (a == "--grid") { need(2); params.grid_x = atoi(argv[++i]); params.grid_y = atoi(argv[++i]); }
        else if (a == "--min-px-dist") { need(1); params.min_px_dist = atoi(argv[++i]); }
        else if (a == "--win") { need(1); params.win_size = atoi(argv[++i]); }
        else if (a == "--levels") { need(1); params.pyr_levels = atoi(argv[++i]); }
        else if (a == "--gyro-seed") params.use_gyro_seed = true;
        else if (a == "--forward-backward") params.forward_backward = true;
        else if (a == "--reject-outliers") params.reject_outliers = true;
        else if (a == "--no-reject-outliers") params.reject_outliers = false;
        else if (a == "--stereo-extrinsic-seed") params.stereo_extrinsic_seed = true;
        else if (a == "--device") { need(1); device = atoi(argv[++i]); }
        else if (a == "--write-dataset") { need(1); write_folder = argv[++i]; }
        else if (a == "--check-config") { need(1); check_file = argv[++i]; }
        else if (a == "--size") { need(2); syn_w = atoi(argv[++i]); syn_h = atoi(argv[++i]); }
        else if (a == "--frames") { need(1); syn_frames = atoi(argv[++i]); }
        else if (a == "--disparity") { need(1); syn_disp = atoi(argv[++i]); }
        else if (a == "--seed") { need(1); syn_seed = (uint32_t)atoi(argv[++i]); }
        else { usage(); return 2; }
    }
    auto with_slash = [](std::string s) { return !s.empty() && s.back() != '/' ? s + "/" : s; };
    if (!write_folder.empty()) {
        if (syn_w < 32 || syn_h < 32 || syn_frames < 1) { usage(); return 2; }
        return write_dataset(with_slash(write_folder), syn_w, syn_h, syn_frames, syn_disp, syn_seed);
    }
    if (!check_file.empty()) {
        param_file = check_file;
        if (!load_params(params)) return 1;
        params.print_state();
        for (const auto& kv : params.camera_extrinsics_mat) {
            printf("cam_%zu_T_cam_imu:", kv.first);
            for (int i = 0; i < 16; i++) printf(" %.9g", kv.second[i]);
            printf("\n");
        }
        return 0;
    }
    data_folder = with_slash(data_folder);
    if (!config_given) param_file = data_folder + "config.yaml";

    // 1. parameters
    if (!load_params(params)) return 1;
    params.print_trackers();
    params.print_state();
    if (params.num_pts < 1 || params.grid_x < 1 || params.grid_y < 1 || params.min_px_dist < 1 || params.win_size < 3 ||
        params.win_size > FT_MAX_WIN || params.pyr_levels < 0 || params.pyr_levels >= FT_MAX_LEVELS) {
        std::cout << "ERROR: tracker options out of range" << std::endl;
        return 2;
    }

    // 2. the device context behind the tracker: two cameras, previous + current pyramid each
    int max_w = 0, max_h = 0;
    for (const auto& wh : params.camera_wh) { max_w = std::max(max_w, wh.second.first); max_h = std::max(max_h, wh.second.second); }
    ft_limits lim;
    lim.max_width = max_w; lim.max_height = max_h;
    lim.max_levels = params.pyr_levels + 1;
    lim.n_streams = 2; lim.n_slots = 2;
    lim.max_points = 2 * (params.num_pts + params.grid_x * params.grid_y + 64);
    lim.max_grid_cells = (params.grid_x + 2) * (params.grid_y + 2);
    lim.max_per_cell = params.num_pts / (params.grid_x * params.grid_y) + 1;
    std::unique_ptr<FrontendBackend> backend;
    try {
        backend.reset(ftk_make_backend(device, lim, nullptr));
    } catch (const std::exception& e) {
        std::cout << "ERROR: no backend: " << e.what() << std::endl;
        return 3;
    }
    if (!backend) { std::cout << "ERROR: no backend" << std::endl; return 3; }
    std::cout << "backend: " << ftk_backend_label() << std::endl;

    // 3. the tracker, as src/test_tracking.cpp:229-231
    std::unique_ptr<TrackKLT> klt(new TrackKLT(backend.get(), params, /*numaruco*/ 0, /*stream_base*/ 0));
    trackFEATS = klt.get();
    trackFEATS->set_calibration(params.camera_intrinsics, params.camera_extrinsics_mat);

    // 4. the whole IMU record first (:234-235), then the images (:237)
    std::thread thd_pub_imu(feed_measurement_imu);
    thd_pub_imu.join();
    feed_measurement_stereo();

    // 5. what the run left in the feature database
    FeatureDatabase* db = trackFEATS->get_feature_database();
    size_t n_obs[2] = {0, 0}, longest = 0, stereo = 0;
    for (const auto& kv : db->get_internal_data()) {
        size_t l[2] = {0, 0};
        for (int cam = 0; cam < 2; cam++)
            if (kv.second->timestamps.count((size_t)cam)) l[cam] = kv.second->timestamps.at((size_t)cam).size();
        n_obs[0] += l[0]; n_obs[1] += l[1];
        longest = std::max(longest, std::max(l[0], l[1]));
        if (l[0] && l[1]) stereo++;
    }
    std::cout << "------------" << std::endl;
    printf("database: %zu features (%zu seen by both cameras), observations left %zu right %zu, longest track %zu\n", db->size(), stereo,
           n_obs[0], n_obs[1], longest);
    trackFEATS = nullptr;
    return 0;
}
