// stand-in for <glm/vec3.hpp>: see _pod.hpp
#pragma once
#include "_pod.hpp"
