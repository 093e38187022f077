#!/bin/bash
# tools/check_reference_caller.sh -- build-container-only evidence for "src/map_sim_example.cpp links unchanged".
#
# Runs `g++ -std=c++14 -fsyntax-only` on the reference's OWN caller, IN PLACE (/root/reference/src/map_sim_example.cpp is read,
# never copied), against THIS repo's drop-in header include/dsp_dynamic.h instead of the reference's: every DSPMap member, macro
# and constant the node uses (my_map.update / getOccupancyMapWithFutureStatus / getVoxelPositionFromIndexPublic / the setters,
# MAP_*_VOXEL_NUM, VOXEL_RESOLUTION, VOXEL_NUM, PREDICTION_TIMES, `using namespace std`) must exist with compatible signatures.
#
# The node also includes ROS, PCL and Eigen, none of which exist in this image.  For a type check of the CALLER they are replaced by
# declaration-only stubs written into a temporary directory below (no behaviour, nothing is linked or run): this is NOT a reference
# build, pins nothing about the reference's arithmetic and is no oracle -- the oracle stays "parity unpinned" (DESIGN.md section 6).
# Nothing here runs on the GPU box (/root/reference does not exist there); tests do not call it.
set -euo pipefail
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -f "$REF/src/map_sim_example.cpp" ] || { echo "no reference tree at $REF: nothing to check"; exit 0; }
T=$(mktemp -d /tmp/dspmap_caller_check.XXXXXX)
trap 'rm -rf "$T"' EXIT
mkdir -p "$T"/{ros,Eigen,pcl/common,pcl/filters,pcl_conversions,sensor_msgs,geometry_msgs,visualization_msgs,gazebo_msgs,nav_msgs,std_msgs}

cat > "$T/ros/ros.h" <<'EOF'
#pragma once
#include <string>
#include <vector>
#include <memory>
namespace ros {
struct Time { double toSec() const; static Time now(); };
struct Duration { Duration(double = 0); };
struct Rate { Rate(double); bool sleep(); };
struct Publisher { template <class M> void publish(const M&) const; };
struct Subscriber {};
struct NodeHandle {
    template <class M> Publisher advertise(const std::string&, unsigned, bool = false);
    template <class A> Subscriber subscribe(const std::string&, unsigned, void (*)(A));
};
struct AsyncSpinner { AsyncSpinner(unsigned); void start(); };
void init(int&, char**, const std::string&);
void spinOnce(); void spin(); bool ok(); void waitForShutdown();
}
namespace std_msgs { struct Header { std::string frame_id; ros::Time stamp; unsigned seq; }; struct ColorRGBA { float r, g, b, a; }; }
#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_INFO_THROTTLE(...) ((void)0)
#define ROS_WARN_THROTTLE(...) ((void)0)
EOF
cat > "$T/std_msgs/Float64.h" <<'EOF'
#pragma once
namespace std_msgs { struct Float64 { double data; }; }
EOF
cat > "$T/geometry_msgs/Pose.h" <<'EOF'
#pragma once
#include "ros/ros.h"
namespace geometry_msgs {
struct Point { double x, y, z; }; struct Vector3 { double x, y, z; }; struct Quaternion { double x, y, z, w; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct Twist { Vector3 linear, angular; };
struct TwistStamped { std_msgs::Header header; Twist twist; };
}
EOF
for h in PoseStamped TwistStamped; do echo '#include "geometry_msgs/Pose.h"' > "$T/geometry_msgs/$h.h"; done
cat > "$T/sensor_msgs/PointCloud2.h" <<'EOF'
#pragma once
#include "ros/ros.h"
namespace sensor_msgs { struct PointCloud2 { std_msgs::Header header; unsigned width, height; };
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr; }
EOF
cat > "$T/visualization_msgs/Marker.h" <<'EOF'
#pragma once
#include "geometry_msgs/Pose.h"
namespace visualization_msgs {
struct Marker { enum { ARROW, CUBE, SPHERE, CYLINDER, LINE_STRIP, LINE_LIST, CUBE_LIST, SPHERE_LIST, POINTS }; enum { ADD, MODIFY, DELETE, DELETEALL };
    std_msgs::Header header; std::string ns; int id, type, action; geometry_msgs::Pose pose; geometry_msgs::Vector3 scale;
    std_msgs::ColorRGBA color; ros::Duration lifetime; std::vector<geometry_msgs::Point> points; std::vector<std_msgs::ColorRGBA> colors; };
struct MarkerArray { std::vector<Marker> markers; };
}
EOF
echo '#include "visualization_msgs/Marker.h"' > "$T/visualization_msgs/MarkerArray.h"
cat > "$T/gazebo_msgs/ModelStates.h" <<'EOF'
#pragma once
#include "geometry_msgs/Pose.h"
namespace gazebo_msgs { struct ModelStates { std::vector<std::string> name; std::vector<geometry_msgs::Pose> pose; std::vector<geometry_msgs::Twist> twist; }; }
EOF
cat > "$T/nav_msgs/Odometry.h" <<'EOF'
#pragma once
#include "geometry_msgs/Pose.h"
namespace nav_msgs { struct Odometry { std_msgs::Header header; }; }
EOF
cat > "$T/Eigen/Eigen" <<'EOF'
#pragma once
namespace Eigen {
template <class S> struct Vec3 { Vec3(); Vec3(S, S, S); S& x(); S& y(); S& z(); const S& x() const; const S& y() const; const S& z() const;
    Vec3 operator+(const Vec3&) const; Vec3 operator-(const Vec3&) const; Vec3 operator*(S) const; Vec3 operator/(S) const; S norm() const;
    S& operator()(int); S& operator[](int); };
template <class S> Vec3<S> operator*(S, const Vec3<S>&);
typedef Vec3<double> Vector3d; typedef Vec3<float> Vector3f;
template <class S> struct Quat { Quat(); Quat(S w, S x, S y, S z); S& w(); S& x(); S& y(); S& z();
    const S& w() const; const S& x() const; const S& y() const; const S& z() const;
    Quat operator*(const Quat&) const; Quat inverse() const; Quat conjugate() const; Quat normalized() const; void normalize();
    Quat slerp(S, const Quat&) const; Vec3<S> operator*(const Vec3<S>&) const; };
typedef Quat<float> Quaternionf; typedef Quat<double> Quaterniond;
}
EOF
cat > "$T/pcl/point_types.h" <<'EOF'
#pragma once
#include <vector>
#include <memory>
#include <cstddef>
namespace pcl {
struct PointXYZ { float x, y, z; };
struct PointXYZRGB { float x, y, z; unsigned char r, g, b; };
struct PointXYZINormal { float x, y, z, intensity, normal_x, normal_y, normal_z; };
template <class T> struct PointCloud { typedef std::shared_ptr<PointCloud<T> > Ptr; typedef std::shared_ptr<const PointCloud<T> > ConstPtr;
    std::vector<T> points; unsigned width, height; bool is_dense;
    void push_back(const T&); void clear(); size_t size() const; bool empty() const; T& operator[](size_t); const T& operator[](size_t) const;
    typename std::vector<T>::iterator begin(); typename std::vector<T>::iterator end(); };
}
EOF
echo '#include <pcl/point_types.h>' > "$T/pcl/point_cloud.h"
echo '#include <pcl/point_types.h>' > "$T/pcl/common/transforms.h"
cat > "$T/pcl/filters/voxel_grid.h" <<'EOF'
#pragma once
#include <pcl/point_types.h>
namespace pcl { template <class T> struct VoxelGrid { void setInputCloud(const typename PointCloud<T>::ConstPtr&); void setLeafSize(float, float, float); void filter(PointCloud<T>&); }; }
EOF
cat > "$T/pcl_conversions/pcl_conversions.h" <<'EOF'
#pragma once
#include <pcl/point_types.h>
#include <sensor_msgs/PointCloud2.h>
namespace pcl { template <class T> void fromROSMsg(const sensor_msgs::PointCloud2&, PointCloud<T>&); template <class T> void toROSMsg(const PointCloud<T>&, sensor_msgs::PointCloud2&); }
EOF

echo "[check_reference_caller] g++ -fsyntax-only $REF/src/map_sim_example.cpp against $ROOT/include/dsp_dynamic.h"
# -I order: the repo's include/ first, so that "dsp_dynamic.h" is THIS repo's header; the reference's include/ is not on the path at all
g++ -std=c++14 -fsyntax-only -Wall -Wno-unused-variable -Wno-unused-but-set-variable -Wno-sign-compare -Wno-unused-function \
    -I"$ROOT/include" -I"$T" "$REF/src/map_sim_example.cpp"
echo "[check_reference_caller] OK: the reference's caller type-checks against the drop-in header (DSPMap surface, macros, using namespace std)"
