/* oracle/ref_tracks.cpp -- TEST INFRASTRUCTURE ONLY (compiled into oracle/_ref/libtracksref.so by oracle/Makefile).
 *
 * The REFERENCE'S OWN BundlerApp::ComputeTracks (src/ComputeTracks.cpp:36-313), compiled from where it lies: this translation unit
 * #includes that file verbatim.  Its class context (BundlerApp / BaseApp / ImageData, src/BundlerApp.h, src/BaseApp.h, src/ImageData.h)
 * drags in the image, geometry and option-parsing code of the whole application, so the three headers it includes are shadowed by their
 * include guards and replaced with the few declarations the function touches:
 *   KeypointMatch (src/keys.h:90-108), MatchIndex / AdjListElem / MatchAdjList / MatchTable (src/BaseApp.h:85, 160-330: per-image adjacency
 *   vectors kept sorted by neighbour index with lower_bound inserts), ImageKey / ImageKeyVector / TrackData (src/ImageData.h), and the
 *   members of ImageData / BundlerApp the function reads or writes.
 * Every statement of the track-building loop itself (queue order, neighbour order, img_marked / m_key_flags tests, equal_range lookups,
 * the >= 2 projections rule, track numbering) is the reference's code.  MakeMatchListsSymmetric (src/MatchTracks.cpp:337-392), which
 * Bundler runs right before ComputeTracks (src/BundlerGeometry.cpp:149-155), is restated in ref_compute_tracks below.
 */
#include <vector>
#include <queue>
#include <algorithm>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cassert>
using namespace std;

#define __keys_h__            /* src/keys.h */
#define __bundlerapp_h__      /* src/BundlerApp.h */
#define ___bundle_util_h___   /* src/BundleUtil.h */

class KeypointMatch {
public:
    KeypointMatch() {}
    KeypointMatch(int idx1, int idx2) : m_idx1(idx1), m_idx2(idx2) {}
    int m_idx1, m_idx2;
};

typedef std::pair<unsigned long, unsigned long> MatchIndex;

class AdjListElem {
public:
    bool operator<(const AdjListElem &other) const { return m_index < other.m_index; }
    unsigned int m_index;
    std::vector<KeypointMatch> m_match_list;
};
typedef std::vector<AdjListElem> MatchAdjList;

class MatchTable {
public:
    MatchTable() {}
    MatchTable(int num_images) { m_match_lists.resize(num_images); }
    void SetMatch(MatchIndex idx) {
        if (Contains(idx)) return;
        AdjListElem e; e.m_index = idx.second;
        MatchAdjList &l = m_match_lists[idx.first];
        MatchAdjList::iterator p = lower_bound(l.begin(), l.end(), e);
        l.insert(p, e);
    }
    void AddMatch(MatchIndex idx, KeypointMatch m) { GetMatchList(idx).push_back(m); }
    void ClearMatch(MatchIndex idx) { if (Contains(idx)) GetMatchList(idx).clear(); }
    std::vector<KeypointMatch> &GetMatchList(MatchIndex idx) {
        AdjListElem e; e.m_index = idx.second;
        MatchAdjList &l = m_match_lists[idx.first];
        std::pair<MatchAdjList::iterator, MatchAdjList::iterator> p = equal_range(l.begin(), l.end(), e);
        assert(p.first != p.second);
        return (p.first)->m_match_list;
    }
    bool Contains(MatchIndex idx) const {
        AdjListElem e; e.m_index = idx.second;
        const MatchAdjList &l = m_match_lists[idx.first];
        return binary_search(l.begin(), l.end(), e);
    }
    void RemoveAll() { for (size_t i = 0; i < m_match_lists.size(); i++) m_match_lists[i].clear(); }
    unsigned int GetNumNeighbors(unsigned int i) { return (unsigned int) m_match_lists[i].size(); }
    MatchAdjList &GetNeighbors(unsigned int i) { return m_match_lists[i]; }
    MatchAdjList::iterator Begin(unsigned int i) { return m_match_lists[i].begin(); }
    MatchAdjList::iterator End(unsigned int i) { return m_match_lists[i].end(); }
    std::vector<MatchAdjList> m_match_lists;
};

typedef std::pair<int, int> ImageKey;
typedef std::vector<ImageKey> ImageKeyVector;

class TrackData {
public:
    TrackData() {}
    TrackData(ImageKeyVector views) : m_views(views) {}
    ImageKeyVector m_views;
};

class ImageData {
public:
    int GetNumKeys() { return m_num_keys; }
    int m_num_keys;
    std::vector<bool> m_key_flags;
    std::vector<int> m_visible_points, m_visible_keys;
};

class BundlerApp {
public:
    void ComputeTracks(int new_image_start);
    int GetNumImages() { return (int) m_image_data.size(); }
    MatchIndex GetMatchIndex(int i1, int i2) { return MatchIndex((unsigned long) i1, (unsigned long) i2); }
    void RemoveAllMatches() { m_matches.RemoveAll(); }
    MatchTable m_matches;
    std::vector<ImageData> m_image_data;
    std::vector<TrackData> m_track_data;
};

#include "ComputeTracks.cpp"      /* resolved through -I$(REF)/src : the reference's file, verbatim */

/* pairs (pair_i[p] < pair_j[p]) with their match lists (idx in image pair_i, idx in image pair_j), as LoadMatchTable stores them
 * (src/BundleIO.cpp:112-166); returns the number of tracks, or -1 when an output buffer is too small. */
extern "C" int ref_compute_tracks(int num_images, const int *num_keys, int num_pairs, const int *pair_i, const int *pair_j,
                                  const int *match_ptr, const int *matches, int new_image_start,
                                  int *track_ptr, int *views, int max_tracks, int max_views)
{
    BundlerApp app;
    app.m_matches = MatchTable(num_images);
    app.m_image_data.resize(num_images);
    for (int i = 0; i < num_images; i++) app.m_image_data[i].m_num_keys = num_keys[i];
    for (int p = 0; p < num_pairs; p++) {
        MatchIndex idx = app.GetMatchIndex(pair_i[p], pair_j[p]);
        app.m_matches.SetMatch(idx);
        std::vector<KeypointMatch> &l = app.m_matches.GetMatchList(idx);
        for (int q = match_ptr[p]; q < match_ptr[p + 1]; q++) l.push_back(KeypointMatch(matches[2 * q], matches[2 * q + 1]));
    }
    /* MakeMatchListsSymmetric, src/MatchTracks.cpp:337-392 */
    for (int i = 0; i < num_images; i++) {
        std::vector<unsigned int> nbrs;
        for (MatchAdjList::iterator it = app.m_matches.Begin(i); it != app.m_matches.End(i); it++) nbrs.push_back(it->m_index);
        for (size_t t = 0; t < nbrs.size(); t++) {
            unsigned int j = nbrs[t];
            if ((int) j <= i) continue;
            const std::vector<KeypointMatch> list = app.m_matches.GetMatchList(app.GetMatchIndex(i, j));
            MatchIndex rev = app.GetMatchIndex(j, i);
            app.m_matches.SetMatch(rev);
            app.m_matches.ClearMatch(rev);
            for (size_t k = 0; k < list.size(); k++) app.m_matches.AddMatch(rev, KeypointMatch(list[k].m_idx2, list[k].m_idx1));
        }
    }
    app.ComputeTracks(new_image_start);
    int nt = (int) app.m_track_data.size(), nv = 0;
    if (nt > max_tracks) return -1;
    for (int t = 0; t < nt; t++) {
        track_ptr[t] = nv;
        const ImageKeyVector &v = app.m_track_data[t].m_views;
        if (nv + (int) v.size() > max_views) return -1;
        for (size_t q = 0; q < v.size(); q++) { views[2 * nv] = v[q].first; views[2 * nv + 1] = v[q].second; nv++; }
    }
    track_ptr[nt] = nv;
    return nt;
}
