#pragma once
#include <memory>
namespace std_msgs { struct Bool { bool data = false; }; typedef std::shared_ptr<Bool> BoolPtr; typedef std::shared_ptr<const Bool> BoolConstPtr; }
