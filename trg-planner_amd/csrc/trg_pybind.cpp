// trg_pybind.cpp -- pybind11 module `trg_planner._trg_pybind`: the reference's Python surface for
// the graph engine (python/trg_planner/pybind/trg_planner_pybind.cpp:23-43: TRG, Edge, NodeState,
// Node) bound over include/trg_shim.hpp, i.e. over the C ABI of libtrg_engine.so.
//
//   TRG(isVerbose, expand_dist, robot_size, sample_num, height_threshold, collision_threshold,
//       update_collision_threshold, safety_factor, goal_tolerance)       pybind :23-24
//   TRG.getGraphCopy(type) -> dict[int, Node]                             pybind :25
//   Edge(dst_id, weight, dist) with rw fields dst_id / weight / dist     pybind :27-31
//   NodeState.{Valid, Invalid, Frontier}                                  pybind :33-36
//   Node(id, pos2d, z, state) with rw fields id / pos / state / edges    pybind :38-43
//
// Eigen is not in this toolchain: Eigen::Vector2f/3f arguments are float32 sequences / numpy arrays
// (what pybind11/eigen.h would convert from and to), `pos` is a writable float32[3] view of the
// node's own storage (pybind11/eigen.h hands out the same for def_readwrite on a Vector3f).
// The rest of class TRG's public surface (trg.h:62-98) is bound too; the reference leaves it
// unbound, TRGPlanner (trg_planner/planner.py) drives the engine through it.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "../../include/trg_shim.hpp"

namespace py = pybind11;
using trg_amd::TRG;
using trg_amd::Vec2f;
using trg_amd::Vec3f;
using farray = py::array_t<float, py::array::c_style | py::array::forcecast>;

namespace {

template <size_t N>
std::array<float, N> vec_from(const py::object &o, const char *what) {
  farray a = farray::ensure(o);
  if (!a || (size_t)a.size() < N)
    throw py::value_error(std::string(what) + ": expected " + std::to_string(N) + " floats");
  std::array<float, N> v;
  const float *p = a.data();
  for (size_t i = 0; i < N; ++i) v[i] = p[i];
  return v;
}

// (n, >=3) float32 cloud -> pointer, count, stride (floats)
struct Cloud {
  farray a;
  size_t n = 0, stride = 3;
  explicit Cloud(const py::object &o) {
    a = farray::ensure(o);
    if (!a) throw py::value_error("point cloud: expected an (n, 3) float array");
    if (a.ndim() == 2 && a.shape(1) >= 3) {
      n = (size_t)a.shape(0);
      stride = (size_t)a.shape(1);
    } else if (a.ndim() == 1 && a.size() % 3 == 0) {
      n = (size_t)a.size() / 3;
    } else if (a.size() == 0) {
      n = 0;
    } else {
      throw py::value_error("point cloud: expected an (n, 3) float array");
    }
  }
  const float *data() const { return a.data(); }
};

py::array_t<float> path_array(const std::vector<Vec3f> &p) {
  py::array_t<float> out({(py::ssize_t)p.size(), (py::ssize_t)3});
  float *d = out.mutable_data();
  for (size_t i = 0; i < p.size(); ++i)
    for (int k = 0; k < 3; ++k) d[3 * i + k] = p[i][k];
  return out;
}

std::vector<Vec3f> path_from(const py::object &o) {
  farray a = farray::ensure(o);
  if (!a || a.size() % 3 != 0) throw py::value_error("path: expected an (n, 3) float array");
  std::vector<Vec3f> p((size_t)a.size() / 3);
  const float *d = a.data();
  for (size_t i = 0; i < p.size(); ++i) p[i] = Vec3f{d[3 * i], d[3 * i + 1], d[3 * i + 2]};
  return p;
}

template <typename T>
py::array_t<T> copy_array(const T *src, py::ssize_t n) {
  py::array_t<T> out(n);
  if (n > 0) std::memcpy(out.mutable_data(), src, (size_t)n * sizeof(T));
  return out;
}

}  // namespace

PYBIND11_MODULE(_trg_pybind, m) {
  m.doc() = "pybind11 bindings of the MI355X-native TRG engine (surface of the reference's trg_planner module)";
  m.attr("__version__") = "1.0.0";

  py::class_<TRG::Edge, std::shared_ptr<TRG::Edge>>(m, "Edge")
      .def(py::init<int, float, float>(), py::arg("dst_id"), py::arg("weight"), py::arg("dist"))
      .def_readwrite("dst_id", &TRG::Edge::dst_id_)
      .def_readwrite("weight", &TRG::Edge::weight_)
      .def_readwrite("dist", &TRG::Edge::dist_)
      .def("__repr__", [](const TRG::Edge &e) {
        return "Edge(dst_id=" + std::to_string(e.dst_id_) + ", weight=" + std::to_string(e.weight_) +
               ", dist=" + std::to_string(e.dist_) + ")";
      });

  py::enum_<TRG::NodeState>(m, "NodeState")
      .value("Valid", TRG::NodeState::Valid)
      .value("Invalid", TRG::NodeState::Invalid)
      .value("Frontier", TRG::NodeState::Frontier);

  py::class_<TRG::Node, std::shared_ptr<TRG::Node>>(m, "Node")
      .def(py::init([](int id, const py::object &pos2d, float z, TRG::NodeState state) {
             return std::make_shared<TRG::Node>(id, vec_from<2>(pos2d, "pos2d"), z, state);
           }),
           py::arg("id"), py::arg("pos2d"), py::arg("z"), py::arg("state"))
      .def_readwrite("id", &TRG::Node::id_)
      .def_property(
          "pos",
          [](py::object self) {  // float32[3] view of the node's own storage, kept alive by the node
            TRG::Node &n = self.cast<TRG::Node &>();
            return py::array_t<float>({(py::ssize_t)3}, {(py::ssize_t)sizeof(float)}, n.pos_.data(), self);
          },
          [](TRG::Node &n, const py::object &v) { n.pos_ = vec_from<3>(v, "pos"); })
      .def_readwrite("state", &TRG::Node::state_)
      .def_readwrite("edges", &TRG::Node::edges_)
      .def("__repr__", [](const TRG::Node &n) {
        return "Node(id=" + std::to_string(n.id_) + ", pos=[" + std::to_string(n.pos_[0]) + ", " +
               std::to_string(n.pos_[1]) + ", " + std::to_string(n.pos_[2]) +
               "], state=" + std::to_string((int)n.state_) + ", deg=" + std::to_string(n.edges_.size()) + ")";
      });

  py::class_<TRG, std::shared_ptr<TRG>>(m, "TRG")
      .def(py::init<bool, float, float, int, float, float, float, float, float, int>(),
           py::arg("isVerbose"), py::arg("expand_dist"), py::arg("robot_size"), py::arg("sample_num"),
           py::arg("height_threshold"), py::arg("collision_threshold"),
           py::arg("update_collision_threshold"), py::arg("safety_factor"), py::arg("goal_tolerance"),
           py::arg("device") = 0)
      .def("getGraphCopy", &TRG::getGraphCopy, "Get the graph", py::arg("type") = "global")
      .def("getGraph", &TRG::getGraph, py::arg("type") = "global")
      .def("lockGraph", &TRG::lockGraph)
      .def("unlockGraph", &TRG::unlockGraph)
      .def("setSampler", &TRG::setSampler, py::arg("seed") = 1, py::arg("table_bits") = 16)
      .def("initGraph",
           [](TRG &t, bool isPreMap, const py::object &start3d) {
             const Vec3f s = vec_from<3>(start3d, "start3d");
             py::gil_scoped_release nogil;
             t.initGraph(isPreMap, s);
           },
           py::arg("isPreMap") = true, py::arg("start3d") = py::make_tuple(0.0f, 0.0f, 0.0f))
      .def("updateGraph", [](TRG &t) {
        py::gil_scoped_release nogil;
        t.updateGraph();
      })
      .def("loadPrebuiltGraph", &TRG::loadPrebuiltGraph, py::arg("filepath"))
      .def("saveGraph", &TRG::saveGraph, py::arg("filepath"))
      .def("setGlobalMap",
           [](TRG &t, const py::object &cloud) {
             Cloud c(cloud);
             py::gil_scoped_release nogil;
             t.setGlobalMap(c.data(), c.n, c.stride);
           },
           py::arg("cloud_xyz"))
      .def("setLocalMap",
           [](TRG &t, const py::object &start2d, const py::object &cloud) {
             const Vec2f s = vec_from<2>(start2d, "start2d");
             Cloud c(cloud);
             py::gil_scoped_release nogil;
             t.setLocalMap(s, c.data(), c.n, c.stride);
           },
           py::arg("start2d"), py::arg("cloud_xyz"))
      .def("voxelFilter",
           [](TRG &t, const py::object &cloud, float leaf) {
             // the pcl::VoxelGrid step of TRGPlanner::loadPrebuiltMap (trg_planner.cpp:91-94)
             Cloud c(cloud);
             std::vector<float> out(3 * c.n + 3);
             size_t n_out = 0;
             int32_t passthrough = 0;
             const TrgStatus st = trg_engine_voxel_filter(t.engine(), c.data(), c.n, c.stride, leaf,
                                                          out.data(), &n_out, &passthrough);
             if (st != TRG_OK) throw std::runtime_error(trg_engine_last_error(t.engine()));
             py::array_t<float> a({(py::ssize_t)n_out, (py::ssize_t)3});
             if (n_out) std::memcpy(a.mutable_data(), out.data(), n_out * 3 * sizeof(float));
             return a;
           },
           py::arg("cloud_xyz"), py::arg("leaf"))
      .def("resetGraph", &TRG::resetGraph, py::arg("type") = "global")
      .def("resetMap", &TRG::resetMap, py::arg("type") = "global")
      .def("isCollision",
           [](TRG &t, const py::object &pos2d, const std::string &type, float threshold) {
             Vec2f p = vec_from<2>(pos2d, "pos2d");
             return t.isCollision(p, type, threshold);
           },
           py::arg("pos2d"), py::arg("type") = "global", py::arg("threshold") = 0.1f)
      .def("isFrontier",
           [](TRG &t, const py::object &pos2d) {
             Vec2f p = vec_from<2>(pos2d, "pos2d");
             return t.isFrontier(p);
           },
           py::arg("pos2d"))
      .def("planSafePath",
           [](TRG &t, const py::object &start2d, const py::object &goal_pose) {
             // (found, out_path[n,3], direct_dist, path_length, avg_risk): the reference's bool return
             // plus its four reference out-parameters (trg.cpp:603-608)
             Vec2f s = vec_from<2>(start2d, "start2d");
             Vec3f g = vec_from<3>(goal_pose, "goal_pose");
             std::vector<Vec3f> path;
             float direct = 0, length = 0, risk = 0;
             bool found;
             {
               py::gil_scoped_release nogil;
               found = t.planSafePath(s, g, path, direct, length, risk);
             }
             return py::make_tuple(found, path_array(path), direct, length, risk);
           },
           py::arg("start2d"), py::arg("goal_pose"))
      .def("checkReadched",  // (sic) the reference's spelling, trg.h:78
           [](TRG &t, const py::object &pos2d) {
             Vec2f p = vec_from<2>(pos2d, "pos2d");
             return t.checkReadched(p);
           },
           py::arg("pos2d"))
      .def("checkReplan",
           [](TRG &t, const py::object &pos2d, const py::object &path) {
             Vec2f p = vec_from<2>(pos2d, "pos2d");
             std::vector<Vec3f> pp = path_from(path);
             return t.checkReplan(p, pp);
           },
           py::arg("pos2d"), py::arg("path"))
      .def("refinePath",
           [](TRG &t, const py::object &in_path) {
             std::vector<Vec3f> in = path_from(in_path), out;
             t.refinePath(in, out);
             return path_array(out);
           },
           py::arg("in_path"))
      .def("getGraphCSR",
           [](TRG &t, const std::string &type) {
             // flat copies of the CSR view: (xyz[V,3], state, rowptr, col, weight, dist, creation_id)
             const TrgCsrView v = t.getGraphCSR(type);
             py::array_t<float> xyz({(py::ssize_t)v.num_nodes, (py::ssize_t)3});
             if (v.num_nodes) std::memcpy(xyz.mutable_data(), v.node_xyz, (size_t)v.num_nodes * 12);
             return py::make_tuple(xyz, copy_array(v.node_state, v.num_nodes),
                                   copy_array(v.rowptr, v.num_nodes + 1), copy_array(v.col, v.num_edges),
                                   copy_array(v.weight, v.num_edges), copy_array(v.dist, v.num_edges),
                                   copy_array(v.creation_id, v.num_nodes));
           },
           py::arg("type") = "global");
}
