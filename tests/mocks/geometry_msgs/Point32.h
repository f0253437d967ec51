#pragma once
namespace geometry_msgs { struct Point32 { float x, y, z; }; }
