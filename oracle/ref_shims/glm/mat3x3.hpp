// stand-in for <glm/mat3x3.hpp>: see _pod.hpp
#pragma once
#include "_pod.hpp"
