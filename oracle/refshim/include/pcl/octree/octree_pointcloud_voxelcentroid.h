/* TEST INFRASTRUCTURE: stand-in header, see refshim/pcl.h */
#pragma once
#include "refshim/pcl.h"
