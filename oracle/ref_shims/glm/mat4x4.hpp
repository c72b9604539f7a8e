// stand-in for <glm/mat4x4.hpp>: see _pod.hpp
#pragma once
#include "_pod.hpp"
