// utilities.h -- constants and helpers feeding the hot path's inputs
// (reference src/utilities.h:12-15, src/utilities.cpp:65-112).
#pragma once
#include <istream>
#include <string>
#include <vector>

#include "linalg.h"

#define PI                3.1415926535897932384626422832795028841971f
#define TWO_PI            6.2831853071795864769252867665590057683943f
#define SQRT_OF_ONE_THIRD 0.5773502691896257645091487805019574556476f
#define EPSILON           0.00001f

namespace utilityCore {
// translation * (Rx * Ry * Rz) * scale, angles in degrees (reference src/utilities.cpp:65-72)
lin::mat4 buildTransformationMatrix(lin::vec3 translation, lin::vec3 rotation, lin::vec3 scale);
// whitespace tokenizer (reference src/utilities.cpp:74-80)
std::vector<std::string> tokenizeString(const std::string &str);
// getline that accepts \n, \r\n and \r and a missing final newline (reference src/utilities.cpp:82-112)
std::istream &safeGetline(std::istream &is, std::string &t);
}  // namespace utilityCore
