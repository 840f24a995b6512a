// API-subset mock (see ../../README.md)
#pragma once
namespace pcl {
double rad2deg(double alpha);
}
