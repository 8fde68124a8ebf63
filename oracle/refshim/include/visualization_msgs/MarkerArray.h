/* TEST INFRASTRUCTURE: stand-in header, see refshim/ros.h */
#pragma once
#include "refshim/ros.h"
