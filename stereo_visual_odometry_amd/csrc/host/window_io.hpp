// window_io.hpp -- a bundle-adjustment window on disk: everything VO::Optimizer::optimize reads out of the
// map (reference: src/solve/optimizer.cpp:19-52) and nothing else, so that windows captured at the call site
// src/frontend.cpp:73 on a machine with the full frontend can be replayed where only the solver runs.  It is
// the successor of the reference's only data fixture format, test/ba_data.txt (scene poses + landmarks,
// parsed at test/ba_test.cpp:30-85), which cannot hold observations, constant flags or intrinsics.
//
// Text form (line oriented, doubles printed with 17 significant digits = exact round trip):
//   hipba-window 1 text
//   camera <z|x> <fx> <fy> <cx> <cy>          z: optimizer.hpp:56-57, x: test/ba_test.h:205-206
//   poses <n>
//   <const 0|1> qx qy qz qw tx ty tz          n lines, Sophus::SE3d::data() order
//   points <n>
//   <const 0|1> x y z
//   observations <n>
//   <pose index> <point index> <u> <v>
//   end
// Binary form: the line "hipba-window 1 binary\n", then little-endian
//   int32 model, double fx fy cx cy, int32 n_poses, int32 n_points, int64 n_obs,
//   double poses[7 n_poses], uint8 pose_const[n_poses], double points[3 n_points], uint8 point_const[n_points],
//   int32 obs_pose[n_obs], int32 obs_point[n_obs], double uv[2 n_obs].
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "hipba.h"

namespace VO {

// a window that owns its arrays; `problem` points into them
struct WindowData {
  std::vector<double> poses, points, uv;
  std::vector<int32_t> obs_pose, obs_point;
  std::vector<uint8_t> pose_const, point_const;
  hipba_problem problem{};

  void bind() {
    problem.n_poses = (int32_t)(poses.size() / 7);
    problem.poses = poses.data();
    problem.pose_const = pose_const.data();
    problem.n_points = (int32_t)(points.size() / 3);
    problem.points = points.data();
    problem.point_const = point_const.data();
    problem.n_obs = (int64_t)obs_pose.size();
    problem.obs_pose = obs_pose.data();
    problem.obs_point = obs_point.data();
    problem.obs_uv = uv.data();
  }
};

namespace detail {
struct File {
  FILE* f;
  File(const std::string& path, const char* mode) : f(fopen(path.c_str(), mode)) {
    if (!f) throw std::runtime_error("cannot open " + path);
  }
  ~File() { if (f) fclose(f); }
  File(const File&) = delete;
  File& operator=(const File&) = delete;
};
template <class T>
void put(FILE* f, const T* p, size_t n) {
  if (n && fwrite(p, sizeof(T), n, f) != n) throw std::runtime_error("short write");
}
template <class T>
void get(FILE* f, T* p, size_t n) {
  if (n && fread(p, sizeof(T), n, f) != n) throw std::runtime_error("truncated window file");
}
}  // namespace detail

inline void writeWindow(const std::string& path, const hipba_problem& p, bool binary = false) {
  detail::File file(path, binary ? "wb" : "w");
  FILE* f = file.f;
  const int64_t n_obs = p.n_obs;
  if (binary) {
    fputs("hipba-window 1 binary\n", f);
    const int32_t model = p.camera_model, np = p.n_poses, nq = p.n_points;
    const double k[4] = {p.fx, p.fy, p.cx, p.cy};
    detail::put(f, &model, 1);
    detail::put(f, k, 4);
    detail::put(f, &np, 1);
    detail::put(f, &nq, 1);
    detail::put(f, &n_obs, 1);
    detail::put(f, p.poses, 7 * (size_t)np);
    std::vector<uint8_t> pc(np, 0), qc(nq, 0);
    if (p.pose_const) memcpy(pc.data(), p.pose_const, np);
    if (p.point_const) memcpy(qc.data(), p.point_const, nq);
    detail::put(f, pc.data(), pc.size());
    detail::put(f, p.points, 3 * (size_t)nq);
    detail::put(f, qc.data(), qc.size());
    detail::put(f, p.obs_pose, (size_t)n_obs);
    detail::put(f, p.obs_point, (size_t)n_obs);
    detail::put(f, p.obs_uv, 2 * (size_t)n_obs);
    return;
  }
  fputs("hipba-window 1 text\n", f);
  fprintf(f, "camera %c %.17g %.17g %.17g %.17g\n", p.camera_model == HIPBA_CAMERA_X_FORWARD ? 'x' : 'z', p.fx, p.fy,
          p.cx, p.cy);
  fprintf(f, "poses %d\n", p.n_poses);
  for (int i = 0; i < p.n_poses; ++i) {
    const double* q = p.poses + 7 * i;
    fprintf(f, "%d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", p.pose_const ? p.pose_const[i] : 0, q[0], q[1], q[2],
            q[3], q[4], q[5], q[6]);
  }
  fprintf(f, "points %d\n", p.n_points);
  for (int i = 0; i < p.n_points; ++i) {
    const double* q = p.points + 3 * i;
    fprintf(f, "%d %.17g %.17g %.17g\n", p.point_const ? p.point_const[i] : 0, q[0], q[1], q[2]);
  }
  fprintf(f, "observations %lld\n", (long long)n_obs);
  for (int64_t o = 0; o < n_obs; ++o)
    fprintf(f, "%d %d %.17g %.17g\n", p.obs_pose[o], p.obs_point[o], p.obs_uv[2 * o], p.obs_uv[2 * o + 1]);
  fputs("end\n", f);
  if (ferror(f)) throw std::runtime_error("write error on " + path);
}

inline void readWindow(const std::string& path, WindowData* w) {
  detail::File file(path, "rb");
  FILE* f = file.f;
  char header[64] = {0};
  if (!fgets(header, sizeof header, f)) throw std::runtime_error("empty window file " + path);
  int version = 0;
  char kind[16] = {0};
  if (sscanf(header, "hipba-window %d %15s", &version, kind) != 2 || version != 1)
    throw std::runtime_error("not a hipba-window version 1 file: " + path);
  hipba_problem& p = w->problem;
  if (!strcmp(kind, "binary")) {
    int32_t model = 0, np = 0, nq = 0;
    int64_t no = 0;
    double k[4];
    detail::get(f, &model, 1);
    detail::get(f, k, 4);
    detail::get(f, &np, 1);
    detail::get(f, &nq, 1);
    detail::get(f, &no, 1);
    if (np < 0 || nq < 0 || no < 0) throw std::runtime_error("negative count in " + path);
    w->poses.resize(7 * (size_t)np); w->pose_const.resize(np);
    w->points.resize(3 * (size_t)nq); w->point_const.resize(nq);
    w->obs_pose.resize(no); w->obs_point.resize(no); w->uv.resize(2 * (size_t)no);
    detail::get(f, w->poses.data(), w->poses.size());
    detail::get(f, w->pose_const.data(), w->pose_const.size());
    detail::get(f, w->points.data(), w->points.size());
    detail::get(f, w->point_const.data(), w->point_const.size());
    detail::get(f, w->obs_pose.data(), w->obs_pose.size());
    detail::get(f, w->obs_point.data(), w->obs_point.size());
    detail::get(f, w->uv.data(), w->uv.size());
    p.camera_model = model;
    p.fx = k[0]; p.fy = k[1]; p.cx = k[2]; p.cy = k[3];
  } else if (!strcmp(kind, "text")) {
    char model = 'z';
    if (fscanf(f, " camera %c %lf %lf %lf %lf", &model, &p.fx, &p.fy, &p.cx, &p.cy) != 5)
      throw std::runtime_error("bad camera line in " + path);
    if (model != 'z' && model != 'x') throw std::runtime_error("camera model must be z or x in " + path);
    p.camera_model = model == 'x' ? HIPBA_CAMERA_X_FORWARD : HIPBA_CAMERA_Z_FORWARD;
    int np = 0, nq = 0;
    long long no = 0;
    if (fscanf(f, " poses %d", &np) != 1 || np < 0) throw std::runtime_error("bad poses line in " + path);
    w->poses.resize(7 * (size_t)np); w->pose_const.resize(np);
    for (int i = 0; i < np; ++i) {
      int c = 0;
      double* q = w->poses.data() + 7 * i;
      if (fscanf(f, "%d %lf %lf %lf %lf %lf %lf %lf", &c, q, q + 1, q + 2, q + 3, q + 4, q + 5, q + 6) != 8)
        throw std::runtime_error("bad pose record in " + path);
      w->pose_const[i] = c ? 1 : 0;
    }
    if (fscanf(f, " points %d", &nq) != 1 || nq < 0) throw std::runtime_error("bad points line in " + path);
    w->points.resize(3 * (size_t)nq); w->point_const.resize(nq);
    for (int i = 0; i < nq; ++i) {
      int c = 0;
      double* q = w->points.data() + 3 * i;
      if (fscanf(f, "%d %lf %lf %lf", &c, q, q + 1, q + 2) != 4) throw std::runtime_error("bad point record in " + path);
      w->point_const[i] = c ? 1 : 0;
    }
    if (fscanf(f, " observations %lld", &no) != 1 || no < 0)
      throw std::runtime_error("bad observations line in " + path);
    w->obs_pose.resize(no); w->obs_point.resize(no); w->uv.resize(2 * (size_t)no);
    for (long long o = 0; o < no; ++o)
      if (fscanf(f, "%d %d %lf %lf", &w->obs_pose[o], &w->obs_point[o], &w->uv[2 * o], &w->uv[2 * o + 1]) != 4)
        throw std::runtime_error("bad observation record in " + path);
  } else {
    throw std::runtime_error("unknown window encoding '" + std::string(kind) + "' in " + path);
  }
  w->bind();
  for (int64_t o = 0; o < p.n_obs; ++o)
    if (p.obs_pose[o] < 0 || p.obs_pose[o] >= p.n_poses || p.obs_point[o] < 0 || p.obs_point[o] >= p.n_points)
      throw std::runtime_error("observation index out of range in " + path);
}

}  // namespace VO
