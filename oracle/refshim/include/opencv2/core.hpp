/* TEST INFRASTRUCTURE: stand-in header, see refshim/cv.h */
#pragma once
#include "refshim/cv.h"
