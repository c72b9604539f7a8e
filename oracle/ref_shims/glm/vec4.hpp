// stand-in for <glm/vec4.hpp>: see _pod.hpp
#pragma once
#include "_pod.hpp"
