// Parameter blocks live in the public C header.
#pragma once
#include "common.h"
