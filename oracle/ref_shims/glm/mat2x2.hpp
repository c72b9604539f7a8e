// stand-in for <glm/mat2x2.hpp>: see _pod.hpp
#pragma once
#include "_pod.hpp"
