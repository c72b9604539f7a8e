// stand-in for <glm/ext/quaternion_float.hpp>: see _pod.hpp
#pragma once
#include "../_pod.hpp"
