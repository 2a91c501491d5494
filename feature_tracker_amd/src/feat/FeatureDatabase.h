// Track database: id -> Feature (call surface of src/feat/FeatureDatabase.h:53-327).  The reference keeps one map under
// one mutex; here the map may be split into shards by id (id % shards), each with its own mutex, memory pool and id
// cache, so that several host threads can write the observations of one large frame at the same time.  One shard (the
// default) is the reference's layout.
// Features handed out with remove=true become the caller's to delete, as in the reference: they are moved out of the
// database's memory pool onto the global heap first, so they may outlive the database.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "ChunkResource.h"
#include "Feature.h"

namespace feature_tracker {

// cache-line aligned: databases of different streams are updated by different host threads
class alignas(64) FeatureDatabase {
public:
    FeatureDatabase() { shards.emplace_back(new Shard()); }
    FeatureDatabase(const FeatureDatabase&) = delete;
    FeatureDatabase& operator=(const FeatureDatabase&) = delete;
    ~FeatureDatabase() = default;

    // Split the (still empty) database into n shards, 1 <= n <= 16.  Returns false, and changes nothing, once features exist.
    bool set_shards(int n) {
        if (n < 1 || n > 16) return false;
        for (auto& s : shards) {
            std::lock_guard<std::mutex> lck(s->mtx);
            if (!s->features_idlookup.empty()) return false;
        }
        shards.clear();
        for (int i = 0; i < n; i++) shards.emplace_back(new Shard());
        return true;
    }
    int shard_count() const { return (int)shards.size(); }
    // which of `parts` writer threads owns an id (see update_features)
    int owner_of(size_t id, int parts) const { return (int)((id % shards.size()) % (size_t)parts); }

    Feature* get_feature(size_t id, bool remove = false) {
        Shard& S = shard_of(id);
        std::lock_guard<std::mutex> lck(S.mtx);
        auto it = S.features_idlookup.find(id);
        if (it == S.features_idlookup.end()) return nullptr;
        Feature* f = it->second;
        if (remove) { S.features_idlookup.erase(it); S.forget_cached_ids(); f = detach(f); }
        return f;
    }

    // append one observation; creates the feature the first time its id is seen (FeatureDatabase.h:96-123)
    void update_feature(size_t id, double timestamp, size_t cam_id, float u, float v, float u_n, float v_n) {
        Shard& S = shard_of(id);
        std::lock_guard<std::mutex> lck(S.mtx);
        Feature*& f = S.features_idlookup[id];
        if (!f) {
            f = new Feature(&S.pool);
            f->featid = id;
        }
        f->append(cam_id, u, v, u_n, v_n, timestamp);
    }

    // the same for n observations of one frame, one lock per shard: uvn holds (u, v, u_n, v_n) per observation.
    // The work is bound by cache misses (id lookup, feature object, its list headers, the tails of its three arrays), so
    // it runs in passes that prefetch ahead: a direct-mapped id cache in front of the map, then the objects, the
    // headers, the tails.
    // (part, parts): only the observations whose id this caller owns (owner_of(id, parts) == part) are written, so
    // `parts` threads may call this at the same time with the same arrays and part = 0 .. parts - 1; with more than
    // one shard they then work side by side.
    void update_features(size_t n, const size_t* ids, double timestamp, size_t cam_id, const float* uvn, int part = 0, int parts = 1) {
        const size_t K = shards.size();
        for (size_t k = 0; k < K; k++) {
            if ((int)(k % (size_t)parts) != part) continue;
            Shard& S = *shards[k];
            std::lock_guard<std::mutex> lck(S.mtx);
            S.scratch.clear();
            S.which.clear();
            // the cache covers the ids of one frame with room to spare (they are recent, nearly consecutive numbers)
            const size_t expect = n / K + 1;
            if (2 * expect > S.id_cache.size() && S.id_cache.size() < ID_CACHE_MAX) {
                size_t want = S.id_cache.size();
                while (want < 4 * expect && want < ID_CACHE_MAX) want *= 2;
                S.id_cache.assign(want, IdSlot());
            }
            const size_t mask = S.id_cache.size() - 1;
            for (size_t i = 0; i < n; i++) {
                if (K > 1 && ids[i] % K != k) continue;
                IdSlot& slot = S.id_cache[(ids[i] / K) & mask];
                if (slot.f == nullptr || slot.id != ids[i]) {
                    Feature*& f = S.features_idlookup[ids[i]];
                    if (!f) {
                        f = new Feature(&S.pool);
                        f->featid = ids[i];
                    }
                    slot.id = ids[i]; slot.f = f;
                }
                S.scratch.push_back(slot.f);
                S.which.push_back((uint32_t)i);
                __builtin_prefetch(slot.f, 1);
            }
            const size_t m = S.scratch.size();
            for (size_t j = 0; j < m; j++) S.scratch[j]->prefetch_lists(cam_id);
            for (size_t j = 0; j < m; j++) S.scratch[j]->prefetch_tail(cam_id);
            for (size_t j = 0; j < m; j++) {
                const float* o = uvn + 4 * (size_t)S.which[j];
                S.scratch[j]->append(cam_id, o[0], o[1], o[2], o[3], timestamp);
            }
        }
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
        for (auto& sp : shards) {
            Shard& S = *sp;
            std::lock_guard<std::mutex> lck(S.mtx);
            S.forget_cached_ids();
            for (auto it = S.features_idlookup.begin(); it != S.features_idlookup.end();) {
                if (it->second->to_delete) { delete it->second; it = S.features_idlookup.erase(it); }
                else ++it;
            }
        }
    }
    // drop measurements at or before timestamp and features left without any (FeatureDatabase.h:283-301)
    void cleanup_measurements(double timestamp) {
        for (auto& sp : shards) {
            Shard& S = *sp;
            std::lock_guard<std::mutex> lck(S.mtx);
            S.forget_cached_ids();
            for (auto it = S.features_idlookup.begin(); it != S.features_idlookup.end();) {
                it->second->clean_older_measurements(timestamp);
                if (it->second->num_measurements() < 1) { delete it->second; it = S.features_idlookup.erase(it); }
                else ++it;
            }
        }
    }

    size_t size() {
        size_t n = 0;
        for (auto& sp : shards) {
            std::lock_guard<std::mutex> lck(sp->mtx);
            n += sp->features_idlookup.size();
        }
        return n;
    }
    std::unordered_map<size_t, Feature*> get_internal_data() {
        std::unordered_map<size_t, Feature*> all;
        for (auto& sp : shards) {
            std::lock_guard<std::mutex> lck(sp->mtx);
            all.insert(sp->features_idlookup.begin(), sp->features_idlookup.end());
        }
        return all;
    }
    // re-key a feature (used by TrackBase::change_feat_id)
    void rekey(size_t id_old, size_t id_new) {
        Shard& A = shard_of(id_old);
        Shard& B = shard_of(id_new);
        if (&A == &B) {
            std::lock_guard<std::mutex> lck(A.mtx);
            auto it = A.features_idlookup.find(id_old);
            if (it == A.features_idlookup.end()) return;
            Feature* f = it->second;
            A.features_idlookup.erase(it);
            A.forget_cached_ids();
            f->featid = id_new;
            // insert semantics, as src/track/TrackBase.h:264: a feature already stored under the new id stays, and the
            // re-keyed one is then unreachable (the reference leaks it; here it is destroyed)
            if (!A.features_idlookup.emplace(id_new, f).second) delete f;
            return;
        }
        // the new id belongs to another shard: the feature moves into that shard's pool (a feature's lists grow under its
        // shard's lock, from its shard's pool)
        std::scoped_lock lck(A.mtx, B.mtx);
        auto it = A.features_idlookup.find(id_old);
        if (it == A.features_idlookup.end()) return;
        Feature* moved = new Feature(*it->second, &B.pool);
        delete it->second;
        A.features_idlookup.erase(it);
        A.forget_cached_ids();
        B.forget_cached_ids();
        moved->featid = id_new;
        if (!B.features_idlookup.emplace(id_new, moved).second) delete moved;
    }

protected:
    static constexpr size_t ID_CACHE = 1024, ID_CACHE_MAX = 65536;   // powers of two: initial size, and the size it may grow to
    struct IdSlot { size_t id = 0; Feature* f = nullptr; };

    struct alignas(64) Shard {
        std::mutex mtx;
        // observation storage of this shard's features: a size-class pool (reuses what shrinking or deleted features
        // give back) on top of huge-page chunks; declared before the map so that it outlives the features on destruction
        ChunkResource chunks;
        std::pmr::unsynchronized_pool_resource pool{std::pmr::pool_options{0, 64 * 1024}, &chunks};
        std::unordered_map<size_t, Feature*> features_idlookup;
        // direct-mapped cache id -> feature in front of the map: a stream's live ids are recent, nearly consecutive numbers
        std::vector<IdSlot> id_cache = std::vector<IdSlot>(ID_CACHE);
        std::vector<Feature*> scratch;
        std::vector<uint32_t> which;
        ~Shard() {
            for (auto& kv : features_idlookup) delete kv.second;
        }
        // every path that removes or re-keys a feature calls this (with mtx held)
        void forget_cached_ids() {
            for (IdSlot& s : id_cache) s = IdSlot();
        }
    };

    Shard& shard_of(size_t id) { return *shards[id % shards.size()]; }

    std::vector<Feature*> collect(bool remove, const std::function<bool(const Feature&)>& pred) {
        std::vector<Feature*> out;
        for (auto& sp : shards) {
            Shard& S = *sp;
            std::lock_guard<std::mutex> lck(S.mtx);
            if (remove) S.forget_cached_ids();
            for (auto it = S.features_idlookup.begin(); it != S.features_idlookup.end();) {
                if (pred(*it->second)) {
                    if (remove) { out.push_back(detach(it->second)); it = S.features_idlookup.erase(it); continue; }
                    out.push_back(it->second);
                }
                ++it;
            }
        }
        return out;
    }

    // copy of a feature on the global heap; the pooled original is destroyed
    static Feature* detach(Feature* pooled) {
        Feature* f = new Feature(*pooled, std::pmr::get_default_resource());
        delete pooled;
        return f;
    }

    std::vector<std::unique_ptr<Shard>> shards;
};

}  // namespace feature_tracker
