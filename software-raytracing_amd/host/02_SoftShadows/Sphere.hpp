#pragma once
#include "../01_Basic/Sphere.hpp"  // identical to 01's in the reference
