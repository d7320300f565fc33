#pragma once
#include "../01_Basic/Types.h"     // the reference's 02 Types.h is identical to 01's
