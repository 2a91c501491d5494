// Track database: id -> Feature under one mutex (call surface of src/feat/FeatureDatabase.h:53-327).
// Features handed out with remove=true become the caller's to delete, as in the reference: they are moved out of the
// database's memory pool onto the global heap first, so they may outlive the database.
#pragma once
#include <functional>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "ChunkResource.h"
#include "Feature.h"

namespace feature_tracker {

// cache-line aligned: databases of different streams are updated by different host threads
class alignas(64) FeatureDatabase {
public:
    FeatureDatabase() = default;
    FeatureDatabase(const FeatureDatabase&) = delete;
    FeatureDatabase& operator=(const FeatureDatabase&) = delete;
    ~FeatureDatabase() {
        for (auto& kv : features_idlookup) delete kv.second;
    }

    Feature* get_feature(size_t id, bool remove = false) {
        std::lock_guard<std::mutex> lck(mtx);
        auto it = features_idlookup.find(id);
        if (it == features_idlookup.end()) return nullptr;
        Feature* f = it->second;
        if (remove) { features_idlookup.erase(it); forget_cached_ids(); f = detach(f); }
        return f;
    }

    // append one observation; creates the feature the first time its id is seen (FeatureDatabase.h:96-123)
    void update_feature(size_t id, double timestamp, size_t cam_id, float u, float v, float u_n, float v_n) {
        std::lock_guard<std::mutex> lck(mtx);
        Feature*& f = features_idlookup[id];
        if (!f) {
            f = new Feature(&pool);
            f->featid = id;
        }
        f->append(cam_id, u, v, u_n, v_n, timestamp);
    }

    // the same for n observations of one frame under a single lock: uvn holds (u, v, u_n, v_n) per observation.
    // The work is bound by cache misses (id lookup, feature object, the tails of its three arrays), so it runs in
    // passes that prefetch ahead: a small direct-mapped id cache in front of the map, then the objects, then the tails.
    void update_features(size_t n, const size_t* ids, double timestamp, size_t cam_id, const float* uvn) {
        std::lock_guard<std::mutex> lck(mtx);
        scratch.resize(n);
        for (size_t i = 0; i < n; i++) {
            IdSlot& slot = id_cache[ids[i] & (ID_CACHE - 1)];
            if (slot.f == nullptr || slot.id != ids[i]) {
                Feature*& f = features_idlookup[ids[i]];
                if (!f) {
                    f = new Feature(&pool);
                    f->featid = ids[i];
                }
                slot.id = ids[i]; slot.f = f;
            }
            scratch[i] = slot.f;
            __builtin_prefetch(slot.f, 1);
        }
        for (size_t i = 0; i < n; i++) scratch[i]->prefetch_tail(cam_id);
        for (size_t i = 0; i < n; i++) scratch[i]->append(cam_id, uvn[4 * i], uvn[4 * i + 1], uvn[4 * i + 2], uvn[4 * i + 3], timestamp);
    }

    // features whose newest measurement in every camera is older than timestamp (FeatureDatabase.h:133-166)
    std::vector<Feature*> features_not_containing_newer(double timestamp, bool remove = false) {
        return collect(remove, [&](const Feature& f) {
            for (const auto& kv : f.timestamps)
                if (!kv.second.empty() && kv.second.back() >= timestamp) return false;
            return true;
        });
    }
    // features with a first measurement before timestamp in some camera (FeatureDatabase.h:175-207)
    std::vector<Feature*> features_containing_older(double timestamp, bool remove = false) {
        return collect(remove, [&](const Feature& f) {
            for (const auto& kv : f.timestamps)
                if (!kv.second.empty() && kv.second.front() < timestamp) return true;
            return false;
        });
    }
    // features with a measurement exactly at timestamp (FeatureDatabase.h:215-255)
    std::vector<Feature*> features_containing(double timestamp, bool remove = false) {
        return collect(remove, [&](const Feature& f) {
            for (const auto& kv : f.timestamps)
                for (double t : kv.second)
                    if (t == timestamp) return true;
            return false;
        });
    }

    // delete features flagged to_delete (FeatureDatabase.h:262-278)
    void cleanup() {
        std::lock_guard<std::mutex> lck(mtx);
        forget_cached_ids();
        for (auto it = features_idlookup.begin(); it != features_idlookup.end();) {
            if (it->second->to_delete) { delete it->second; it = features_idlookup.erase(it); }
            else ++it;
        }
    }
    // drop measurements at or before timestamp and features left without any (FeatureDatabase.h:283-301)
    void cleanup_measurements(double timestamp) {
        std::lock_guard<std::mutex> lck(mtx);
        forget_cached_ids();
        for (auto it = features_idlookup.begin(); it != features_idlookup.end();) {
            it->second->clean_older_measurements(timestamp);
            if (it->second->num_measurements() < 1) { delete it->second; it = features_idlookup.erase(it); }
            else ++it;
        }
    }

    size_t size() {
        std::lock_guard<std::mutex> lck(mtx);
        return features_idlookup.size();
    }
    std::unordered_map<size_t, Feature*> get_internal_data() {
        std::lock_guard<std::mutex> lck(mtx);
        return features_idlookup;
    }
    // re-key a feature (used by TrackBase::change_feat_id)
    void rekey(size_t id_old, size_t id_new) {
        std::lock_guard<std::mutex> lck(mtx);
        auto it = features_idlookup.find(id_old);
        if (it == features_idlookup.end()) return;
        Feature* f = it->second;
        features_idlookup.erase(it);
        forget_cached_ids();
        f->featid = id_new;
        features_idlookup[id_new] = f;
    }

protected:
    std::vector<Feature*> collect(bool remove, const std::function<bool(const Feature&)>& pred) {
        std::vector<Feature*> out;
        std::lock_guard<std::mutex> lck(mtx);
        if (remove) forget_cached_ids();
        for (auto it = features_idlookup.begin(); it != features_idlookup.end();) {
            if (pred(*it->second)) {
                if (remove) { out.push_back(detach(it->second)); it = features_idlookup.erase(it); continue; }
                out.push_back(it->second);
            }
            ++it;
        }
        return out;
    }

    // copy of a feature on the global heap; the pooled original is destroyed
    static Feature* detach(Feature* pooled) {
        Feature* f = new Feature(*pooled, std::pmr::get_default_resource());
        delete pooled;
        return f;
    }

    // every path that removes or re-keys a feature calls this (with mtx held)
    void forget_cached_ids() {
        for (IdSlot& s : id_cache) s = IdSlot();
    }

    std::mutex mtx;
    // observation storage of this database's features: a size-class pool (reuses what shrinking or deleted features
    // give back) on top of huge-page chunks; declared before the map so that it outlives the features on destruction
    ChunkResource chunks;
    std::pmr::unsynchronized_pool_resource pool{std::pmr::pool_options{0, 64 * 1024}, &chunks};
    std::unordered_map<size_t, Feature*> features_idlookup;
    // direct-mapped cache id -> feature in front of the map: a stream's live ids are recent, nearly consecutive numbers
    static constexpr size_t ID_CACHE = 1024;   // power of two
    struct IdSlot { size_t id = 0; Feature* f = nullptr; };
    std::vector<IdSlot> id_cache = std::vector<IdSlot>(ID_CACHE);
    std::vector<Feature*> scratch;
};

}  // namespace feature_tracker
