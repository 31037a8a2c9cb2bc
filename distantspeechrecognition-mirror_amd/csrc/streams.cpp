#include "common.h"
