// stand-in for <glm/ext/quaternion_transform.hpp>: see _pod.hpp
#pragma once
#include "../_pod.hpp"
