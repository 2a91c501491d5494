// One tracked landmark: per-camera raw and normalized observations with their timestamps
// (mirrors the members of src/feat/Feature.h:39-90 that the tracker fills).
#pragma once
#include <algorithm>
#include <array>
#include <cstddef>
#include <memory_resource>
#include <unordered_map>
#include <vector>

namespace feature_tracker {

class Feature {
public:
    typedef std::pmr::vector<std::array<float, 2>> UvList;
    typedef std::pmr::vector<double> TimeList;

    // storage comes from `mem` (a feature database passes its own pool; the default is the global heap)
    explicit Feature(std::pmr::memory_resource* mem = std::pmr::get_default_resource()) : uvs(mem), uvs_norm(mem), timestamps(mem) {}
    Feature(const Feature& o, std::pmr::memory_resource* mem)
        : featid(o.featid), to_delete(o.to_delete), uvs(o.uvs, mem), uvs_norm(o.uvs_norm, mem), timestamps(o.timestamps, mem),
          anchor_cam_id(o.anchor_cam_id), anchor_clone_timestamp(o.anchor_clone_timestamp), p_FinA(o.p_FinA), p_FinG(o.p_FinG) {}
    Feature(const Feature&) = delete;
    Feature& operator=(const Feature&) = delete;

    size_t featid = 0;
    bool to_delete = false;
    std::pmr::unordered_map<size_t, UvList> uvs;        // raw pixels per camera
    std::pmr::unordered_map<size_t, UvList> uvs_norm;   // undistorted, normalized
    std::pmr::unordered_map<size_t, TimeList> timestamps;
    int anchor_cam_id = -1;
    double anchor_clone_timestamp = 0.0;
    std::array<double, 3> p_FinA{{0, 0, 0}};
    std::array<double, 3> p_FinG{{0, 0, 0}};

    // append one observation; the three per-camera vectors of the camera used last are remembered, so a track that
    // stays in one camera costs three push_backs and no hash lookups (map nodes never move)
    void append(size_t cam_id, float u, float v, float u_n, float v_n, double timestamp) {
        if (!hot_ts || hot_cam != cam_id) {
            hot_cam = cam_id;
            hot_uv = &uvs[cam_id]; hot_uvn = &uvs_norm[cam_id]; hot_ts = &timestamps[cam_id];
            if (hot_ts->capacity() < 16) { hot_uv->reserve(16); hot_uvn->reserve(16); hot_ts->reserve(16); }
        }
        hot_uv->push_back({{u, v}});
        hot_uvn->push_back({{u_n, v_n}});
        hot_ts->push_back(timestamp);
    }

    // pull the three list headers that prefetch_tail and append read (they live in the maps' nodes)
    void prefetch_lists(size_t cam_id) const {
        if (!hot_ts || hot_cam != cam_id) return;
        __builtin_prefetch(hot_uv, 1);
        __builtin_prefetch(hot_uvn, 1);
        __builtin_prefetch(hot_ts, 1);
    }

    // pull the cache lines the next append(cam_id, ...) will write (no effect on the contents)
    void prefetch_tail(size_t cam_id) const {
        if (!hot_ts || hot_cam != cam_id) return;
        __builtin_prefetch(hot_uv->data() + hot_uv->size(), 1);
        __builtin_prefetch(hot_uvn->data() + hot_uvn->size(), 1);
        __builtin_prefetch(hot_ts->data() + hot_ts->size(), 1);
    }

    // keep only measurements taken at one of valid_times (src/feat/Feature.cpp:29-58)
    void clean_old_measurements(const std::vector<double>& valid_times) {
        filter([&](double t) { return std::find(valid_times.begin(), valid_times.end(), t) != valid_times.end(); });
    }
    // drop measurements at or before timestamp (src/feat/Feature.cpp:60-89)
    void clean_older_measurements(double timestamp) {
        filter([&](double t) { return t > timestamp; });
    }
    size_t num_measurements() const {
        size_t n = 0;
        for (const auto& kv : timestamps) n += kv.second.size();
        return n;
    }

private:
    size_t hot_cam = 0;
    UvList* hot_uv = nullptr;
    UvList* hot_uvn = nullptr;
    TimeList* hot_ts = nullptr;

    template <class Keep>
    void filter(Keep keep) {
        for (auto& kv : timestamps) {
            auto& ts = kv.second;
            auto& a = uvs[kv.first];
            auto& b = uvs_norm[kv.first];
            size_t o = 0;
            for (size_t i = 0; i < ts.size(); i++) {
                if (!keep(ts[i])) continue;
                ts[o] = ts[i]; a[o] = a[i]; b[o] = b[i];
                o++;
            }
            ts.resize(o); a.resize(o); b.resize(o);
        }
    }
};

}  // namespace feature_tracker
