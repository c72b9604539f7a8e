// stand-in for <glm/vec2.hpp>: see _pod.hpp
#pragma once
#include "_pod.hpp"
