// ORACLE — TEST INFRASTRUCTURE ONLY (see ba_oracle.hpp header).
//
// CPU restatement of the Voxel-SLAM voxel hash map + adaptive octree used by local
// mapping: VOXEL_LOC/hash (TL:24-49), pointVar/Plane/Bf_var (VM:18-121), SlideWindow
// (VM:1009-1042), OctoTree (VM:1047-1881: push, push_fix, plane_judge, allocate,
// allocate_fix, fix_divide, subdivide, plane_update, recut, margi, tras_opt),
// cut_voxel / cut_voxel_multi / cut_voxel(fix) (VM:1896-2152), multi_recut / multi_margi
// (voxelslam.cpp VS:1590-1737) and the ring-map rotation (VS:2014-2019).
// The reference's process-wide globals (VM:98-104, VM:1046) become MapConfig + mp here.
// Worker threads of the reference only partition independent root voxels; this
// restatement runs them sequentially in the same map iteration order.
#pragma once
#include "ba_oracle.hpp"
#include <unordered_map>
#include <cstdint>

namespace vso {

struct pointVar { V3 pnt; M3 var; };  // VM:18-34
typedef std::vector<pointVar> PVec;

struct Plane {  // VM:83-96
  V3 center, normal;
  M6 plane_var;
  float radius = 0;
  bool is_plane = false;
};

struct MapConfig {  // VM:98-104 + LocalBA params (VS:917-931)
  int win_size = 10;
  double voxel_size = 1.0;
  int max_layer = 2;
  int max_points = 100;
  double min_eigen_value = 0.0025;
  double plane_eigen_value_thre[4] = {0.25, 0.25, 0.25, 0.25};  // already inverted (VS:930-931)
  double min_point[4] = {5, 5, 5, 5};
  int thread_num = 5;
};

struct VOXEL_LOC {  // TL:24-35
  int64_t x, y, z;
  VOXEL_LOC(int64_t vx = 0, int64_t vy = 0, int64_t vz = 0) : x(vx), y(vy), z(vz) {}
  bool operator==(const VOXEL_LOC &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct VoxelLocHash {  // TL:39-48 (size_t arithmetic, HASH_P 116101, MAX_N 1e10)
  size_t operator()(const VOXEL_LOC &s) const {
    const uint64_t HASH_P = 116101ull, MAX_N = 10000000000ull;
    return (size_t)((((uint64_t)s.z * HASH_P) % MAX_N + (uint64_t)s.y) * HASH_P) % MAX_N + (uint64_t)s.x;
  }
};

// VM:1907-1918: float narrowing, "-1 if negative", truncation toward zero.
inline VOXEL_LOC voxel_key(const V3 &pw, double voxel_size) {
  float loc[3];
  for (int j = 0; j < 3; j++) {
    loc[j] = pw[j] / voxel_size;
    if (loc[j] < 0) loc[j] -= 1;
  }
  return VOXEL_LOC((int64_t)loc[0], (int64_t)loc[1], (int64_t)loc[2]);
}

inline void Bf_var(const pointVar &pv, Mat<9, 9> &bcov, const V3 &vec) {  // VM:106-121
  Mat<6, 3> Bi;
  Bi(0, 0) = 2 * vec[0];
  Bi(1, 0) = vec[1]; Bi(1, 1) = vec[0];
  Bi(2, 0) = vec[2]; Bi(2, 2) = vec[0];
  Bi(3, 1) = 2 * vec[1];
  Bi(4, 1) = vec[2]; Bi(4, 2) = vec[1];
  Bi(5, 2) = 2 * vec[2];
  Mat<6, 3> Biup = Bi * pv.var;
  bcov.setBlock<6, 6>(0, 0, Biup * Bi.transpose());
  bcov.setBlock<6, 3>(0, 6, Biup);
  bcov.setBlock<3, 6>(6, 0, Biup.transpose());
  bcov.setBlock<3, 3>(6, 6, pv.var);
}

// calcBodyVar voxelslam.hpp:180-200 (float narrowing of range / range_var kept).  DEG2RAD is PCL's macro
// (pcl/pcl_macros.h, PCL 1.10 per README.md:27; the header is not in the reference tree): ((x) * 0.017453293).
inline void calcBodyVar(V3 &pb, const float range_inc, const float degree_inc, M3 &var) {
  if (pb[2] == 0) pb[2] = 0.0001;
  float range = std::sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
  float range_var = range_inc * range_inc;
  const double dv = std::pow(std::sin((degree_inc) * 0.017453293), 2);
  V3 direction = pb / pb.norm();
  M3 direction_hat = hat(direction);
  V3 b1 = v3(1, 1, -(direction[0] + direction[1]) / direction[2]);
  b1 = b1 / b1.norm();
  V3 b2 = cross(b1, direction);
  b2 = b2 / b2.norm();
  Mat<3, 2> N;
  for (int r = 0; r < 3; r++) { N(r, 0) = b1[r]; N(r, 1) = b2[r]; }
  Mat<3, 2> A = (direction_hat * N) * (double)range;
  Mat<2, 2> dvar; dvar(0, 0) = dv; dvar(1, 1) = dv;
  var = (direction * direction.transpose()) * (double)range_var + A * dvar * A.transpose();
}
// var_init voxelslam.hpp:210-234: body covariance, then point and covariance moved by the extrinsic
inline void var_init(const M3 &extR, const V3 &extp, PVec &pv, double dept_err, double beam_err) {
  for (pointVar &q : pv) {
    calcBodyVar(q.pnt, dept_err, beam_err, q.var);
    q.pnt = extR * q.pnt + extp;
    q.var = extR * q.var * extR.transpose();
  }
}
// pvec_update voxelslam.hpp:242-265: var becomes world-frame, pnt stays body-frame, pwld = R p + t
inline void pvec_update(PVec &pv, const IMUST &x, std::vector<V3> &pwld) {
  M3 rot_var = x.cov.block<3, 3>(0, 0), tsl_var = x.cov.block<3, 3>(3, 3);
  for (pointVar &q : pv) {
    M3 phat = hat(q.pnt);
    q.var = x.R * q.var * x.R.transpose() + phat * rot_var * phat.transpose() + tsl_var;
    pwld.push_back(x.R * q.pnt + x.p);
  }
}

struct SlideWindow {  // VM:1009-1042
  std::vector<PVec> points;
  std::vector<PointCluster> pcrs_local;
  explicit SlideWindow(int wdsize) { pcrs_local.resize(wdsize); points.resize(wdsize); }
  void resize(int wdsize) { if ((int)points.size() != wdsize) { points.resize(wdsize); pcrs_local.resize(wdsize); } }
  void clear() { for (size_t i = 0; i < points.size(); i++) { points[i].clear(); pcrs_local[i].clear(); } }
};

struct MapCtx { MapConfig cfg; std::vector<int> mp; };

struct OctoTree {  // VM:1047-1881
  MapCtx *ctx;
  SlideWindow *sw = nullptr;
  PointCluster pcr_add;
  Mat<9, 9> cov_add;
  PointCluster pcr_fix;
  PVec point_fix;
  int layer, octo_state, wdsize;
  OctoTree *leaves[8];
  double voxel_center[3] = {0, 0, 0};
  double jour = 0;
  float quater_length = 0;
  Plane plane;
  bool isexist = false;
  V3 eig_value; M3 eig_vector;
  int last_num = 0, opt_state = -1;

  OctoTree(MapCtx *c, int _l, int _w) : ctx(c), layer(_l), octo_state(0), wdsize(_w) { for (int i = 0; i < 8; i++) leaves[i] = nullptr; }
  ~OctoTree() { for (int i = 0; i < 8; i++) delete leaves[i]; }

  void push(int ord, const pointVar &pv, const V3 &pw, std::vector<SlideWindow *> &sws) {  // VM:1105-1143
    if (sw == nullptr) {
      if (sws.size() != 0) { sw = sws.back(); sws.pop_back(); sw->resize(wdsize); }
      else sw = new SlideWindow(wdsize);
    }
    if (!isexist) isexist = true;
    int mord = ctx->mp[ord];
    if (layer < ctx->cfg.max_layer) sw->points[mord].push_back(pv);
    sw->pcrs_local[mord].push(pv.pnt);
    pcr_add.push(pw);
    Mat<9, 9> Bi; Bf_var(pv, Bi, pw);
    cov_add += Bi;
  }
  void push_fix(pointVar &pv) {  // VM:1149-1162
    if (layer < ctx->cfg.max_layer) point_fix.push_back(pv);
    pcr_fix.push(pv.pnt);
    pcr_add.push(pv.pnt);
    Mat<9, 9> Bi; Bf_var(pv, Bi, pv.pnt);
    cov_add += Bi;
  }
  void push_fix_novar(pointVar &pv) {  // VM:1168-1178
    if (layer < ctx->cfg.max_layer) point_fix.push_back(pv);
    pcr_fix.push(pv.pnt);
    pcr_add.push(pv.pnt);
  }
  bool plane_judge(V3 &ev) {  // VM:1185-1195
    return (ev[0] < ctx->cfg.min_eigen_value && (ev[0] / ev[2]) < ctx->cfg.plane_eigen_value_thre[layer]);
  }
  OctoTree *child_for(const V3 &p) {  // VM:1214-1232 (shared by allocate/allocate_fix/fix_divide/subdivide)
    int xyz[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) if (p[k] > voxel_center[k]) xyz[k] = 1;
    int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
    if (leaves[leafnum] == nullptr) {
      leaves[leafnum] = new OctoTree(ctx, layer + 1, wdsize);
      leaves[leafnum]->voxel_center[0] = voxel_center[0] + (2 * xyz[0] - 1) * quater_length;
      leaves[leafnum]->voxel_center[1] = voxel_center[1] + (2 * xyz[1] - 1) * quater_length;
      leaves[leafnum]->voxel_center[2] = voxel_center[2] + (2 * xyz[2] - 1) * quater_length;
      leaves[leafnum]->quater_length = quater_length / 2;
    }
    return leaves[leafnum];
  }
  void allocate(int ord, const pointVar &pv, const V3 &pw, std::vector<SlideWindow *> &sws) {  // VM:1204-1237
    if (octo_state == 0) push(ord, pv, pw, sws);
    else child_for(pw)->allocate(ord, pv, pw, sws);
  }
  void allocate_fix(pointVar &pv) {  // VM:1239-1264
    if (octo_state == 0) push_fix_novar(pv);
    else if (layer < ctx->cfg.max_layer) child_for(pv.pnt)->allocate_fix(pv);
  }
  void fix_divide(std::vector<SlideWindow *> &) {  // VM:1270-1299
    for (pointVar &pv : point_fix) child_for(pv.pnt)->push_fix(pv);
  }
  void subdivide(int si, IMUST &xx, std::vector<SlideWindow *> &sws) {  // VM:1307-1338
    for (pointVar &pv : sw->points[ctx->mp[si]]) {
      V3 pw = xx.R * pv.pnt + xx.p;
      child_for(pw)->push(si, pv, pw, sws);
    }
  }
  void plane_update() {  // VM:1344-1388
    plane.center = pcr_add.v / (double)pcr_add.N;
    int l = 0;
    V3 u[3] = {eig_vector.col(0), eig_vector.col(1), eig_vector.col(2)};
    double nv = 1.0 / pcr_add.N;
    Mat<3, 9> u_c;
    for (int k = 0; k < 3; k++)
      if (k != l) {
        M3 ukl = u[k] * u[l].transpose();
        Mat<1, 9> fkl;
        fkl[0] = ukl(0, 0); fkl[1] = ukl(1, 0) + ukl(0, 1); fkl[2] = ukl(2, 0) + ukl(0, 2);
        fkl[3] = ukl(1, 1); fkl[4] = ukl(1, 2) + ukl(2, 1); fkl[5] = ukl(2, 2);
        V3 tail = -(u[l] * dot(u[k], plane.center) + u[k] * dot(u[l], plane.center));
        fkl[6] = tail[0]; fkl[7] = tail[1]; fkl[8] = tail[2];
        u_c += (u[k] * fkl) * (nv / (eig_value[l] - eig_value[k]));
      }
    Mat<3, 9> Jc = u_c * cov_add;
    plane.plane_var.setBlock<3, 3>(0, 0, Jc * u_c.transpose());
    M3 Jc_N = Jc.block<3, 3>(0, 6) * nv;
    plane.plane_var.setBlock<3, 3>(0, 3, Jc_N);
    plane.plane_var.setBlock<3, 3>(3, 0, Jc_N.transpose());
    plane.plane_var.setBlock<3, 3>(3, 3, cov_add.block<3, 3>(6, 6) * (nv * nv));
    plane.normal = u[0];
    plane.radius = eig_value[2];
  }
  void recut(int win_count, std::vector<IMUST> &x_buf, std::vector<SlideWindow *> &sws) {  // VM:1396-1456
    if (octo_state == 0) {
      if (layer >= 0) {
        opt_state = -1;
        if (pcr_add.N <= ctx->cfg.min_point[layer]) { plane.is_plane = false; return; }
        if (!isexist || sw == nullptr) return;
        eig3_sym(pcr_add.cov(), eig_value, eig_vector);
        plane.is_plane = plane_judge(eig_value);
        if (plane.is_plane) return;
        else if (layer >= ctx->cfg.max_layer) return;
      }
      if (pcr_fix.N != 0) { fix_divide(sws); PVec().swap(point_fix); }
      for (int i = 0; i < win_count; i++) subdivide(i, x_buf[i], sws);
      sw->clear(); sws.push_back(sw); sw = nullptr;
      octo_state = 1;
    }
    for (int i = 0; i < 8; i++) if (leaves[i] != nullptr) leaves[i]->recut(win_count, x_buf, sws);
  }
  // VM:1465-1598.  Returns false where the reference would printf+exit(0) (VM:1488-1492).
  bool margi(int win_count, int mgsize, std::vector<IMUST> &x_buf, const LidarFactor &vox_opt) {
    bool ok = true;
    if (octo_state == 0 && layer >= 0) {
      if (!isexist || sw == nullptr) return true;
      std::vector<PointCluster> pcrs_world(wdsize);
      if (opt_state >= int(vox_opt.pcr_adds.size())) return false;
      const std::vector<int> &mp = ctx->mp;
      if (opt_state >= 0) {
        pcr_add = vox_opt.pcr_adds[opt_state];
        eig_value = vox_opt.eig_values[opt_state];
        eig_vector = vox_opt.eig_vectors[opt_state];
        opt_state = -1;
        for (int i = 0; i < mgsize; i++)
          if (sw->pcrs_local[mp[i]].N != 0) pcrs_world[i].transform(sw->pcrs_local[mp[i]], x_buf[i]);
      } else {
        pcr_add = pcr_fix;
        for (int i = 0; i < win_count; i++)
          if (sw->pcrs_local[mp[i]].N != 0) {
            pcrs_world[i].transform(sw->pcrs_local[mp[i]], x_buf[i]);
            pcr_add += pcrs_world[i];
          }
        if (plane.is_plane) eig3_sym(pcr_add.cov(), eig_value, eig_vector);
      }
      if (pcr_fix.N < ctx->cfg.max_points && plane.is_plane)
        if (pcr_add.N - last_num >= 5 || last_num <= 10) { plane_update(); last_num = pcr_add.N; }
      if (pcr_fix.N < ctx->cfg.max_points) {
        for (int i = 0; i < mgsize; i++)
          if (pcrs_world[i].N != 0) {
            pcr_fix += pcrs_world[i];
            for (pointVar pv : sw->points[mp[i]]) { pv.pnt = x_buf[i].R * pv.pnt + x_buf[i].p; point_fix.push_back(pv); }
          }
      } else {
        for (int i = 0; i < mgsize; i++) if (pcrs_world[i].N != 0) pcr_add -= pcrs_world[i];
        if (point_fix.size() != 0) PVec().swap(point_fix);
      }
      for (int i = 0; i < mgsize; i++)
        if (sw->pcrs_local[mp[i]].N != 0) { sw->pcrs_local[mp[i]].clear(); sw->points[mp[i]].clear(); }
      isexist = !(pcr_fix.N >= pcr_add.N);
    } else {
      isexist = false;
      for (int i = 0; i < 8; i++)
        if (leaves[i] != nullptr) {
          ok = leaves[i]->margi(win_count, mgsize, x_buf, vox_opt) && ok;
          isexist = isexist || leaves[i]->isexist;
        }
    }
    return ok;
  }
  void tras_opt(LidarFactor &vox_opt) {  // VM:1605-1638
    if (octo_state == 0) {
      if (layer >= 0 && isexist && plane.is_plane && sw != nullptr) {
        if (eig_value[0] / eig_value[1] > 0.12) return;
        double coe = 1;
        std::vector<PointCluster> pcrs(wdsize);
        for (int i = 0; i < wdsize; i++) pcrs[i] = sw->pcrs_local[ctx->mp[i]];
        opt_state = (int)vox_opt.plvec_voxels.size();
        vox_opt.push_voxel(pcrs, pcr_fix, coe, eig_value, eig_vector, pcr_add);
      }
    } else {
      for (int i = 0; i < 8; i++) if (leaves[i] != nullptr) leaves[i]->tras_opt(vox_opt);
    }
  }
  // VM:1649-1721 (float temporaries at VM:1657-1661 kept)
  int match(const V3 &wld, Plane *&pla, double &max_prob, const M3 &var_wld, double &sigma_d, OctoTree *&oc) {
    int flag = 0;
    if (octo_state == 0) {
      if (plane.is_plane) {
        float dis_to_plane = std::fabs(dot(plane.normal, V3(wld - plane.center)));
        float dis_to_center = V3(plane.center - wld).squaredNorm();
        float range_dis = (dis_to_center - dis_to_plane * dis_to_plane);
        if (range_dis <= 3 * 3 * plane.radius) {
          Mat<1, 6> J_nq;
          V3 d = wld - plane.center;
          for (int k = 0; k < 3; k++) { J_nq[k] = d[k]; J_nq[3 + k] = -plane.normal[k]; }
          double sigma_l = (J_nq * plane.plane_var * J_nq.transpose())[0];
          sigma_l += dot(plane.normal, V3(var_wld * plane.normal));
          if (dis_to_plane < 3 * std::sqrt(sigma_l)) {
            oc = this; sigma_d = sigma_l; pla = &plane;
            flag = 1;
          }
        }
      }
    } else {
      int xyz[3] = {0, 0, 0};
      for (int k = 0; k < 3; k++) if (wld[k] > voxel_center[k]) xyz[k] = 1;
      int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
      if (leaves[leafnum] != nullptr) flag = leaves[leafnum]->match(wld, pla, max_prob, var_wld, sigma_d, oc);
    }
    return flag;
  }
  bool inside(const V3 &wld) const {  // VM:1836-1848
    double hl = quater_length * 2;
    return (wld[0] >= voxel_center[0] - hl && wld[0] <= voxel_center[0] + hl && wld[1] >= voxel_center[1] - hl &&
            wld[1] <= voxel_center[1] + hl && wld[2] >= voxel_center[2] - hl && wld[2] <= voxel_center[2] + hl);
  }
  void clear_slwd(std::vector<SlideWindow *> &sws) {  // VM:1856-1880
    if (octo_state != 0) for (int i = 0; i < 8; i++) if (leaves[i] != nullptr) leaves[i]->clear_slwd(sws);
    if (sw != nullptr) { sw->clear(); sws.push_back(sw); sw = nullptr; }
  }
};

typedef std::unordered_map<VOXEL_LOC, OctoTree *, VoxelLocHash> VoxelHashMap;

struct VoxelMapOracle {
  MapCtx ctx;
  MapConfig &cfg;
  VoxelHashMap surf_map, surf_map_slide;
  std::vector<SlideWindow *> sws;  // the reference keeps thread_num pools (VS:957); pooling has no numeric effect

  explicit VoxelMapOracle(const MapConfig &c) : cfg(ctx.cfg) {
    ctx.cfg = c;
    ctx.mp.resize(c.win_size);
    for (int i = 0; i < c.win_size; i++) ctx.mp[i] = i;  // VS:3158-3160
  }
  ~VoxelMapOracle() {
    for (auto &kv : surf_map) { kv.second->clear_slwd(sws); delete kv.second; }
    for (SlideWindow *s : sws) delete s;
  }
  OctoTree *new_root(const VOXEL_LOC &position) {  // VM:1935-1943
    OctoTree *ot = new OctoTree(&ctx, 0, cfg.win_size);
    ot->voxel_center[0] = (0.5 + position.x) * cfg.voxel_size;
    ot->voxel_center[1] = (0.5 + position.y) * cfg.voxel_size;
    ot->voxel_center[2] = (0.5 + position.z) * cfg.voxel_size;
    ot->quater_length = cfg.voxel_size / 4.0;
    return ot;
  }
  void cut_voxel(PVec &pvec, int win_count, std::vector<V3> &pwld) {  // VM:1896-1949
    for (size_t i = 0; i < pvec.size(); i++) {
      VOXEL_LOC position = voxel_key(pwld[i], cfg.voxel_size);
      auto iter = surf_map.find(position);
      if (iter != surf_map.end()) {
        iter->second->allocate(win_count, pvec[i], pwld[i], sws);
        iter->second->isexist = true;
        if (surf_map_slide.find(position) == surf_map_slide.end()) surf_map_slide[position] = iter->second;
      } else {
        OctoTree *ot = new_root(position);
        ot->allocate(win_count, pvec[i], pwld[i], sws);
        surf_map[position] = ot;
        surf_map_slide[position] = ot;
      }
    }
  }
  void cut_voxel_multi(PVec &pvec, int win_count, std::vector<V3> &pwld) {  // VM:1964-2096
    std::unordered_map<OctoTree *, std::vector<int>> map_pvec;
    for (size_t i = 0; i < pvec.size(); i++) {
      VOXEL_LOC position = voxel_key(pwld[i], cfg.voxel_size);
      auto iter = surf_map.find(position);
      OctoTree *ot = nullptr;
      if (iter != surf_map.end()) {
        iter->second->isexist = true;
        if (surf_map_slide.find(position) == surf_map_slide.end()) surf_map_slide[position] = iter->second;
        ot = iter->second;
      } else {
        ot = new_root(position);
        surf_map[position] = ot;
        surf_map_slide[position] = ot;
      }
      map_pvec[ot].push_back((int)i);
    }
    if ((int)map_pvec.size() < cfg.thread_num) return;  // VM:2044-2045: scan silently dropped
    for (auto &kv : map_pvec)
      for (int k : kv.second) kv.first->allocate(win_count, pvec[k], pwld[k], sws);
  }
  void cut_voxel_fix(PVec &pvec, double jour) {  // VM:2108-2152
    for (pointVar &pv : pvec) {
      VOXEL_LOC position = voxel_key(pv.pnt, cfg.voxel_size);
      auto iter = surf_map.find(position);
      if (iter != surf_map.end()) iter->second->allocate_fix(pv);
      else {
        OctoTree *ot = new_root(position);
        ot->push_fix_novar(pv);
        ot->jour = jour;
        surf_map[position] = ot;
      }
    }
  }
  // multi_recut VS:1682-1737 (multi=true: early return when #roots < thread_num) or the
  // single-thread form of motion_init VS:699-703 (multi=false, iterates surf_map).
  void recut_all(int win_count, std::vector<IMUST> &xs, LidarFactor &voxopt, bool multi) {
    if (multi) {
      if ((int)surf_map_slide.size() < cfg.thread_num) return;
      for (auto &kv : surf_map_slide) kv.second->recut(win_count, xs, sws);
      for (auto &kv : surf_map_slide) kv.second->tras_opt(voxopt);
    } else {
      for (auto &kv : surf_map) { kv.second->recut(win_count, xs, sws); kv.second->tras_opt(voxopt); }
    }
  }
  bool multi_margi(int win_count, std::vector<IMUST> &xs, LidarFactor &voxopt, double jour = 0) {  // VS:1590-1679
    if ((int)surf_map_slide.size() < cfg.thread_num) return true;
    bool ok = true;
    for (auto &kv : surf_map_slide) kv.second->jour = jour;   // VS:1628
    for (auto &kv : surf_map_slide) ok = kv.second->margi(win_count, 1, xs, voxopt) && ok;
    for (auto iter = surf_map_slide.begin(); iter != surf_map_slide.end();) {
      if (iter->second->isexist) iter++;
      else { iter->second->clear_slwd(sws); surf_map_slide.erase(iter++); }
    }
    return ok;
  }
  void prune(double jour, int dist = 700) {  // voxelslam.cpp:1800-1823
    for (auto iter = surf_map.begin(); iter != surf_map.end();) {
      int dis = jour - iter->second->jour;
      if (dis < dist) iter++;
      else {
        surf_map_slide.erase(iter->first);   // (never the case in the reference: sliding-map roots carry a fresh jour)
        iter->second->clear_slwd(sws);
        delete iter->second;
        surf_map.erase(iter++);
      }
    }
  }
  void slide(int mgsize) {  // VS:2014-2019
    for (int i = 0; i < cfg.win_size; i++) { ctx.mp[i] += mgsize; if (ctx.mp[i] >= cfg.win_size) ctx.mp[i] -= cfg.win_size; }
  }

  // match(feat_map, wld, pla, var_wld, sigma_d, oc)  VM:2167-2205
  int match(const V3 &wld, Plane *&pla, const M3 &var_wld, double &sigma_d, OctoTree *&oc) {
    int flag = 0;
    VOXEL_LOC position = voxel_key(wld, cfg.voxel_size);
    auto iter = surf_map.find(position);
    if (iter != surf_map.end()) {
      double max_prob = 0;
      flag = iter->second->match(wld, pla, max_prob, var_wld, sigma_d, oc);
    }
    return flag;
  }

  // bool VOXEL_SLAM::lio_state_estimation(PVecPtr pptr)  voxelslam.cpp:962-1098 — iterated EKF scan-to-map update of
  // x_curr (state + 15x15 cov).  pvec holds body-frame points with body-frame covariance.  trace (optional): per
  // iteration [match_num, |rot_add|, |tra_add|].
  bool lio_state_estimation(const PVec &pvec, IMUST &x_curr, std::vector<double> *trace = nullptr) {
    IMUST x_prop = x_curr;
    const int num_max_iter = 4;
    bool EKF_stop_flg = 0, flg_EKF_converged = 0;
    Mat<15, 15> G, H_T_H, I_STATE = Mat<15, 15>::Identity();
    int rematch_num = 0, match_num = 0;
    int psize = (int)pvec.size();
    std::vector<OctoTree *> octos(psize, nullptr);
    M3 nnt;
    Mat<15, 15> cov_inv = inverse_lu<15>(x_curr.cov);
    for (int iterCount = 0; iterCount < num_max_iter; iterCount++) {
      M6 HTH; V6 HTz;
      M3 rot_var = x_curr.cov.block<3, 3>(0, 0);
      M3 tsl_var = x_curr.cov.block<3, 3>(3, 3);
      match_num = 0;
      nnt.setZero();
      for (int i = 0; i < psize; i++) {
        const pointVar &pv = pvec[i];
        M3 phat = hat(pv.pnt);
        M3 var_world = x_curr.R * pv.var * x_curr.R.transpose() + phat * rot_var * phat.transpose() + tsl_var;
        V3 wld = x_curr.R * pv.pnt + x_curr.p;
        double sigma_d = 0;
        Plane *pla = nullptr;
        int flag = 0;
        if (octos[i] != nullptr && octos[i]->inside(wld)) {
          double max_prob = 0;
          flag = octos[i]->match(wld, pla, max_prob, var_world, sigma_d, octos[i]);
        } else {
          flag = match(wld, pla, var_world, sigma_d, octos[i]);
        }
        if (flag) {
          Plane &pp = *pla;
          double R_inv = 1.0 / (0.0005 + sigma_d);
          double resi = dot(pp.normal, V3(wld - pp.center));
          V6 jac;
          V3 jr3 = phat * x_curr.R.transpose() * pp.normal;
          for (int k = 0; k < 3; k++) { jac[k] = jr3[k]; jac[3 + k] = pp.normal[k]; }
          HTH += (jac * jac.transpose()) * R_inv;
          HTz -= jac * (R_inv * resi);
          nnt += pp.normal * pp.normal.transpose();
          match_num++;
        }
      }
      H_T_H.setZero();  // (the reference only ever writes block (0,0); the rest stays zero)
      H_T_H.setBlock<6, 6>(0, 0, HTH);
      Mat<15, 15> K_1 = inverse_lu<15>(H_T_H + cov_inv);
      Mat<15, 6> K6 = K_1.block<15, 6>(0, 0);
      G.setZero();
      G.setBlock<15, 6>(0, 0, K6 * HTH);
      Mat<15, 1> vec;   // x_prop - x_curr  (IMUST::operator- TL:164-173)
      vec.setBlock<3, 1>(0, 0, Log(x_curr.R.transpose() * x_prop.R));
      vec.setBlock<3, 1>(3, 0, x_prop.p - x_curr.p);
      vec.setBlock<3, 1>(6, 0, x_prop.v - x_curr.v);
      vec.setBlock<3, 1>(9, 0, x_prop.bg - x_curr.bg);
      vec.setBlock<3, 1>(12, 0, x_prop.ba - x_curr.ba);
      Mat<15, 1> solution = K6 * HTz + vec - G.block<15, 6>(0, 0) * vec.block<6, 1>(0, 0);
      // x_curr += solution  (TL:154-162)
      x_curr.R = x_curr.R * Exp(solution.block<3, 1>(0, 0));
      x_curr.p += solution.block<3, 1>(3, 0);
      x_curr.v += solution.block<3, 1>(6, 0);
      x_curr.bg += solution.block<3, 1>(9, 0);
      x_curr.ba += solution.block<3, 1>(12, 0);
      V3 rot_add = solution.block<3, 1>(0, 0), tra_add = solution.block<3, 1>(3, 0);
      if (trace) { trace->push_back(match_num); trace->push_back(rot_add.norm()); trace->push_back(tra_add.norm()); }
      EKF_stop_flg = false;
      flg_EKF_converged = false;
      if ((rot_add.norm() * 57.3 < 0.01) && (tra_add.norm() * 100 < 0.015)) flg_EKF_converged = true;
      if (flg_EKF_converged || ((rematch_num == 0) && (iterCount == num_max_iter - 2))) rematch_num++;
      if (rematch_num >= 2 || (iterCount == num_max_iter - 1)) {
        x_curr.cov = (I_STATE - G) * x_curr.cov;
        EKF_stop_flg = true;
      }
      if (EKF_stop_flg) break;
    }
    V3 evalue; M3 evec;
    eig3_sym(nnt, evalue, evec);
    return !(evalue[0] < 14);
  }

  // ---- test dumps
  void dump_rec(const VOXEL_LOC &k, OctoTree *n, int path, double *&out, int &cnt, int max_leaves) {
    if (n->octo_state == 0) {
      if (cnt < max_leaves) {
        double *o = out;
        o[0] = (double)k.x; o[1] = (double)k.y; o[2] = (double)k.z; o[3] = n->layer; o[4] = path;
        o[5] = n->pcr_add.N; o[6] = n->pcr_fix.N; o[7] = n->plane.is_plane; o[8] = n->isexist; o[9] = n->opt_state;
        for (int i = 0; i < 3; i++) o[10 + i] = n->eig_value[i];
        for (int i = 0; i < 9; i++) o[13 + i] = n->eig_vector[i];
        o[22] = n->pcr_add.P(0, 0); o[23] = n->pcr_add.P(1, 0); o[24] = n->pcr_add.P(2, 0);
        o[25] = n->pcr_add.P(1, 1); o[26] = n->pcr_add.P(2, 1); o[27] = n->pcr_add.P(2, 2);
        o[28] = n->pcr_add.v[0]; o[29] = n->pcr_add.v[1]; o[30] = n->pcr_add.v[2]; o[31] = n->pcr_add.N;
        for (int i = 0; i < 3; i++) { o[32 + i] = n->plane.center[i]; o[35 + i] = n->plane.normal[i]; }
        o[38] = n->plane.radius;
        out += 39;
      }
      cnt++;
    } else {
      for (int i = 0; i < 8; i++)
        if (n->leaves[i]) dump_rec(k, n->leaves[i], n->layer == 0 ? i : path * 8 + i, out, cnt, max_leaves);
    }
  }
  int dump_leaves(double *out, int max_leaves) {
    int cnt = 0;
    for (auto &kv : surf_map) dump_rec(kv.first, kv.second, 0, out, cnt, max_leaves);
    return cnt;
  }
  void dump_pv_rec(OctoTree *n, double *&out, int &cnt, int max_leaves) {
    if (n->octo_state == 0) {
      if (cnt < max_leaves) { for (int i = 0; i < 36; i++) out[i] = n->plane.plane_var[i]; out += 36; }
      cnt++;
    } else for (int i = 0; i < 8; i++) if (n->leaves[i]) dump_pv_rec(n->leaves[i], out, cnt, max_leaves);
  }
  int dump_plane_var(double *out, int max_leaves) {
    int cnt = 0;
    for (auto &kv : surf_map) dump_pv_rec(kv.second, out, cnt, max_leaves);
    return cnt;
  }
  // cov_add (9x9, VM:106-121 summed at VM:1138-1140), upper triangle row by row (45), same traversal order as dump_leaves
  void dump_ca_rec(OctoTree *n, double *&out, int &cnt, int max_leaves) {
    if (n->octo_state == 0) {
      if (cnt < max_leaves) { int k = 0; for (int r = 0; r < 9; r++) for (int c = r; c < 9; c++) out[k++] = n->cov_add(r, c); out += 45; }
      cnt++;
    } else for (int i = 0; i < 8; i++) if (n->leaves[i]) dump_ca_rec(n->leaves[i], out, cnt, max_leaves);
  }
  int dump_cov_add(double *out, int max_leaves) {
    int cnt = 0;
    for (auto &kv : surf_map) dump_ca_rec(kv.second, out, cnt, max_leaves);
    return cnt;
  }
};

}  // namespace vso
