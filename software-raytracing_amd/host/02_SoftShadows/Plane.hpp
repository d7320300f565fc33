#pragma once
#include "../01_Basic/Plane.hpp"   // identical to 01's in the reference
