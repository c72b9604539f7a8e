// stand-in for <glm/gtc/quaternion.hpp>: see _pod.hpp
#pragma once
#include "../_pod.hpp"
