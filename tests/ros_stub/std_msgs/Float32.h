#pragma once
namespace std_msgs { struct Float32 { float data = 0.f; }; }
